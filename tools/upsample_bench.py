"""HIP-event timing of the bilinear x2 kernels (csrc/upsample.hip) on SepConv's shapes at 256 x 448, N = 8: the decoder's full maps and the
sub-networks' window; checks the outputs against torch (forward) and the autograd adjoint (backward).  SAVFI_HIP_LIB selects a variant."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meta_interpolation_amd import hip_ops, _hip
from meta_interpolation_amd.sepconv.model import MetaNetwork

dev = torch.device("cuda")


def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(1e3 * a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2]


net = MetaNetwork()
win = net._window(256, 448, 384, 512)
cy0, cy1, cx0, cx1 = win['crop']
cases = [("decoder 512 @12x16", 8, 512, 12, 16, None), ("decoder 256 @24x32", 8, 256, 24, 32, None), ("decoder 128 @48x64", 8, 128, 48, 64, None),
         ("decoder 64 @96x128", 8, 64, 96, 128, None), ("subnet window 51 @%dx%d -> 258x450" % (cy1 - cy0, cx1 - cx0), 8, 51, cy1 - cy0, cx1 - cx0, win)]
for name, N, C, H, W, w in cases:
    x = torch.randn(N, C, H, W, device=dev, requires_grad=True)
    if w is None:
        f = lambda: hip_ops.upsample_bilinear2x(x, True)
        ref = torch.nn.functional.interpolate(x.detach(), scale_factor=2, mode='bilinear', align_corners=True)
    else:
        f = lambda: hip_ops.upsample_bilinear2x_window(x, w['half'], (w['crop'][0], w['crop'][2]), w['up'], True)
        full = torch.zeros(N, C, w['half'][0], w['half'][1], device=dev)
        full[:, :, cy0:cy1, cx0:cx1] = x.detach()
        u = w['up']
        ref = torch.nn.functional.interpolate(full, scale_factor=2, mode='bilinear', align_corners=True)[:, :, u[0]:u[0] + u[2], u[1]:u[1] + u[3]]
    y = f()
    err = (y.detach() - ref).abs().max().item()
    gy = torch.randn_like(y)
    tf = timeit(lambda: f())
    # backward through the C ABI (the autograd call costs more host time than the small maps' kernels take)
    lib, st = _hip.lib(), _hip.current_stream()
    (gref,) = torch.autograd.grad(f(), x, gy)
    gx = torch.empty_like(x)
    if w is None:
        geo = (H, W, 0, 0, H, W, 0, 0, 2 * H, 2 * W, 1)
    else:
        geo = (w['half'][0], w['half'][1], cy0, cx0, cy1 - cy0, cx1 - cx0, w['up'][0], w['up'][1], w['up'][2], w['up'][3], 1)
    fb = lambda: _hip.check(lib.savfi_upsample2x_window_bwd_f32(gy.data_ptr(), gx.data_ptr(), N * C, *geo, st), "bwd")
    tb = timeit(fb)
    berr = (gx - gref).abs().max().item()
    nbytes = 4 * (x.numel() + y.numel())
    print(json.dumps({"lib": os.path.basename(os.environ.get("SAVFI_HIP_LIB", "default")), "case": name, "fwd_us": round(tf, 1), "fwd_GBps": round(nbytes / tf / 1e3),
                      "bwd_us": round(tb, 1), "bwd_GBps": round(nbytes / tb / 1e3), "fwd_max_err": err, "bwd_vs_autograd": berr}), flush=True)
