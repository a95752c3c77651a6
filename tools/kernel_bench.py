"""Isolated timing of each savfi kernel at the BASELINE shapes (HIP events, many launches).

    python tools/kernel_bench.py [--iters 50] [--only sepconv]

Prints one JSON object per kernel: mean/min launch time, algorithmic bytes, achieved GB/s and the
fraction of the 8 TB/s HBM peak (and, for sepconv, achieved fp32 TFLOP/s against the 157.3 peak).
Used for tuning; bench.py reports the same quantity measured inside the real inner loop.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from meta_interpolation_amd import _hip, hip_ops  # noqa: E402
from meta_interpolation_amd.sepconv.sepconv_op.sepconv import algorithmic_bytes  # noqa: E402

DEV = "cuda"


def timeit(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    us = [1e3 * a.elapsed_time(b) for a, b in evs]
    return sum(us) / len(us), min(us)


def report(name, us_mean, us_min, nbytes, flops=None, **extra):
    d = {"kernel": name, "mean_us": round(us_mean, 2), "min_us": round(us_min, 2), "algorithmic_MB": round(nbytes / 1e6, 2),
         "GBps": round(nbytes / us_mean / 1e3, 1), "hbm_frac": round(nbytes / us_mean / 1e3 / 8000.0, 4)}
    if flops:
        d["TFLOPs"] = round(flops / us_mean / 1e6, 2)
        d["valu_frac"] = round(flops / us_mean / 1e6 / 157.3, 4)
    d.update(extra)
    print(json.dumps(d), flush=True)


def bench_sepconv(iters, B=1, Ho=384, Wo=512, K=51, C=3):
    lib, st = _hip.lib(), _hip.current_stream()
    inp = torch.rand(B, C, Ho + K - 1, Wo + K - 1, device=DEV)
    v = torch.randn(B, K, Ho, Wo, device=DEV) / 7
    h = torch.randn(B, K, Ho, Wo, device=DEV) / 7
    gO = torch.randn(B, C, Ho, Wo, device=DEV)
    out, gV, gH = torch.empty_like(gO), torch.empty_like(v), torch.empty_like(h)
    px = B * Ho * Wo
    f = lambda: _hip.check(lib.savfi_sepconv_fwd_f32(inp.data_ptr(), v.data_ptr(), h.data_ptr(), out.data_ptr(),
                                                     B, C, Ho, Wo, K, st), "fwd")
    m, mn = timeit(f, iters)
    report("sepconv_fwd B=%d %dx%d" % (B, Ho, Wo), m, mn, algorithmic_bytes(B, C, Ho, Wo, K), 2.0 * px * C * (K * K + K))
    b = lambda: _hip.check(lib.savfi_sepconv_bwd_f32(inp.data_ptr(), v.data_ptr(), h.data_ptr(), gO.data_ptr(), None,
                                                     gV.data_ptr(), gH.data_ptr(), B, C, Ho, Wo, K, st), "bwd")
    m, mn = timeit(b, iters)
    report("sepconv_bwd(gV+gH) B=%d %dx%d" % (B, Ho, Wo), m, mn, algorithmic_bytes(B, C, Ho, Wo, K, 2),
           2.0 * px * 2 * C * (K * K + K))


def bench_update(iters, model="sepconv"):
    from tests.helpers import build_plugin
    net = build_plugin(model, DEV)
    ws = [p.detach() for p in net.parameters() if p.requires_grad]
    gs = [torch.randn_like(w) for w in ws]
    P = sum(w.numel() for w in ws)
    lr_s = [torch.tensor(1e-3, device=DEV) for _ in ws]
    lr_e = [torch.full_like(w, 1e-3) for w in ws]
    ms, ss = [torch.zeros_like(w) for w in ws], [torch.zeros_like(w) for w in ws]
    bc1, sb2 = [0.1] * len(ws), [0.1] * len(ws)
    with torch.no_grad():
        m, mn = timeit(lambda: hip_ops.mt_update(_hip.RULE_SGD, _hip.LR_SCALAR, ws, gs, lr_s), iters)
        report("mt_update LSLR-SGD %s (%d tensors, %.1fM)" % (model, len(ws), P / 1e6), m, mn, 12 * P,
               note="wall incl. host ctypes + torch.empty_like per tensor")
        m, mn = timeit(lambda: hip_ops.mt_update(_hip.RULE_SGD, _hip.LR_ELEMENT, ws, gs, lr_e), iters)
        report("mt_update MetaSGD-SGD %s" % model, m, mn, 16 * P)
        m, mn = timeit(lambda: hip_ops.mt_update(_hip.RULE_ADAM, _hip.LR_SCALAR, ws, gs, lr_s, m=ms, s=ss, bc1=bc1,
                                                 sqrt_bc2=sb2), iters)
        report("mt_update LSLR-Adam %s" % model, m, mn, 28 * P)
        m, mn = timeit(lambda: hip_ops.mt_update(_hip.RULE_ADAMAX_MSGD, _hip.LR_ELEMENT, ws, gs, lr_e, bc1=bc1), iters)
        report("mt_update MetaSGD-Adamax %s" % model, m, mn, 16 * P)
        m, mn = timeit(lambda: hip_ops.mt_mean(gs), iters)
        report("mt_mean %s" % model, m, mn, 4 * P)


def bench_misc(iters):
    fr = torch.rand(1, 6, 256, 256, device=DEV) * 2 - 1
    x3 = torch.tanh(torch.randn(1, 3, 256, 256, device=DEV))
    with torch.no_grad():
        m, mn = timeit(lambda: hip_ops.voxel_warp_blend(fr, x3), iters)
    report("voxelwarp_fwd 256x256", m, mn, 4 * 12 * 256 * 256)
    # straight through the C ABI, 20 launches per timed region: the autograd wrapper costs more host time than these kernels run
    from meta_interpolation_amd import _hip
    lib, REP = _hip.lib(), 20
    for (n, h, w) in ((2, 256, 448), (1, 768, 1280)):          # Super SloMo / RRIN: a padded 256x448 pair, padded 720p
        img = torch.rand(n, 3, h, w, device=DEV)
        # a smooth flow field of a few pixels (optical flow is piecewise smooth): low-pass filtered noise
        flow = torch.nn.functional.avg_pool2d(torch.randn(n, 2, h + 32, w + 32, device=DEV) * 40, 33, stride=1).contiguous()
        gout, out, gflow = torch.randn(n, 3, h, w, device=DEV), torch.empty(n, 3, h, w, device=DEV), torch.empty_like(flow)

        def fwd():
            for _ in range(REP):
                lib.savfi_flowwarp_fwd_f32(img.data_ptr(), flow.data_ptr(), out.data_ptr(), n, 3, h, w, _hip.current_stream())

        def bwd():
            for _ in range(REP):
                lib.savfi_flowwarp_bwd_f32(img.data_ptr(), flow.data_ptr(), gout.data_ptr(), gflow.data_ptr(), n, 3, h, w,
                                           _hip.current_stream())
        m, mn = timeit(fwd, iters)
        report("flowwarp_fwd N=%d %dx%d" % (n, h, w), m / REP, mn / REP, 4 * n * h * w * (3 + 2 + 3))
        m, mn = timeit(bwd, iters)
        report("flowwarp_bwd N=%d %dx%d" % (n, h, w), m / REP, mn / REP, 4 * n * h * w * (3 + 2 + 3 + 2))
    x = torch.randn(1, 3, 768, 1280, device=DEV)
    with torch.no_grad():
        m, mn = timeit(lambda: hip_ops.pixel_shuffle(x, 1 / 8), iters)
        report("pixel_unshuffle 768x1280 r8", m, mn, 2 * 4 * 3 * 768 * 1280)
        y = hip_ops.pixel_shuffle(x, 1 / 8)
        m, mn = timeit(lambda: hip_ops.pixel_shuffle(y, 8), iters)
        report("pixel_shuffle 768x1280 r8", m, mn, 2 * 4 * 3 * 768 * 1280)
        a, b = torch.rand(1, 3, 256, 448, device=DEV), torch.rand(1, 3, 256, 448, device=DEV)
        m, mn = timeit(lambda: hip_ops.l1_loss(a, b), iters)
        report("l1_loss 256x448", m, mn, 2 * 4 * 3 * 256 * 448)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--only", default=None)
    ap.add_argument("--hw", default="384x512,256x448", help="sepconv output sizes: padded canvas, frame window")
    ap.add_argument("--batches", default="1,2", help="sepconv batch sizes (8 = 4 tasks in lockstep x the support pair)")
    o = ap.parse_args()
    if o.only in (None, "sepconv"):
        for hw in o.hw.split(","):
            Ho, Wo = (int(t) for t in hw.split("x"))
            for B in (int(t) for t in o.batches.split(",")):
                bench_sepconv(o.iters, B=B, Ho=Ho, Wo=Wo)
    if o.only in (None, "update"):
        bench_update(o.iters, "sepconv")
        bench_update(o.iters, "cain")
    if o.only in (None, "misc"):
        bench_misc(o.iters)
