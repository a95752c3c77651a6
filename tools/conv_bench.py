"""3x3 convolution micro-benchmark: savfi_conv3x3_f32 (Winograd on fp32 MFMA) vs MIOpen through F.conv2d,
forward and data gradient, on the layer shapes of the SepConv network at 256x448 (padded canvas 384x512).

    python tools/conv_bench.py [--iters 20] [--n 2]
"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meta_interpolation_amd import hip_ops  # noqa: E402

# (Ci, Co, H, W) of the distinct 3x3 layers (encoder, decoder, sub-networks on the frame window)
LAYERS = [(6, 32, 384, 512), (32, 32, 384, 512), (32, 64, 192, 256), (64, 64, 192, 256), (64, 128, 96, 128),
          (128, 128, 96, 128), (128, 256, 48, 64), (256, 256, 48, 64), (256, 512, 24, 32), (512, 512, 24, 32),
          (512, 512, 12, 16), (512, 256, 24, 32), (256, 128, 48, 64), (128, 64, 96, 128),
          (64, 64, 136, 233), (64, 51, 136, 233), (51, 51, 258, 450)]


def timeit(fn, iters, reps=10):
    """Device time per call: `reps` calls captured in one hipGraph (no host gaps between the kernels, filter
    transform / layout kernels included), median over `iters` replays."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record()
        g.replay()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 / reps for a, b in ev)
    return ts[len(ts) // 2]


# CAIN at 720p (C5): 127 convolutions 192->192 on the 1/8-resolution 96x160 map; RRIN / Super SloMo U-Net levels at 256x448
EXTRA = {"cain": [(192, 192, 96, 160), (384, 192, 96, 160), (192, 192, 32, 56)],
         "unet": [(32, 32, 256, 448), (64, 64, 128, 224), (128, 128, 64, 112), (256, 256, 32, 56), (512, 512, 16, 28)]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--set", default="sepconv", choices=["sepconv"] + sorted(EXTRA))
    o = ap.parse_args()
    dev = torch.device("cuda")
    for (ci, co, h, w) in (LAYERS if o.set == "sepconv" else EXTRA[o.set]):
        x = torch.randn(o.n, ci, h, w, device=dev)
        wt = torch.randn(co, ci, 3, 3, device=dev) / (3 * ci ** 0.5)
        b = torch.randn(co, device=dev)
        gy = torch.randn(o.n, co, h, w, device=dev)
        gflop = 2.0 * 9 * ci * co * h * w * o.n / 1e9
        ref = F.conv2d(x, wt, b, padding=1)
        got = hip_ops.conv3x3(x, wt, b, 0, 1.0)
        err = float((got - ref).abs().max() / ref.abs().max())
        t_mi_f = timeit(lambda: F.conv2d(x, wt, b, padding=1), o.iters)
        t_my_f = timeit(lambda: hip_ops.conv3x3(x, wt, b, 0, 1.0), o.iters)
        t_mi_b = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, wt, None, [1, 1], [1, 1], [1, 1], False, [0, 0],
                                                                    1, [True, False, False]), o.iters)
        t_my_b = timeit(lambda: hip_ops.conv3x3(gy, wt, None, 1, 1.0), o.iters)
        print(json.dumps({"layer": "%d->%d @%dx%d N=%d" % (ci, co, h, w, o.n), "gflop": round(gflop, 2),
                          "fwd_miopen_us": round(t_mi_f, 1), "fwd_savfi_us": round(t_my_f, 1),
                          "fwd_savfi_TFLOPs_direct_equiv": round(gflop / t_my_f * 1e3, 1), "fwd_ratio": round(t_mi_f / t_my_f, 2), "bwd_ratio": round(t_mi_b / t_my_b, 2),
                          "bwd_miopen_us": round(t_mi_b, 1), "bwd_savfi_us": round(t_my_b, 1),
                          "rel_err_vs_miopen": err}), flush=True)


if __name__ == "__main__":
    main()
