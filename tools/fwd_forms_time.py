"""HIP-event timing of the 3 x 3 forward / data gradient in its two forms on layer shapes of the bench configs: Winograd F(2x2,3x3) on the
fp32 matrix cores (savfi_conv3x3_tasks_pre) against the direct split-bf16 kernel (savfi_convk_tasks_pre).  python tools/fwd_forms_time.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meta_interpolation_amd import hip_ops
SHAPES = [  # N, T, Ci, Co, H, W
    (32, 4, 64, 51, 137, 236), (32, 4, 51, 64, 137, 236), (32, 4, 64, 64, 137, 236), (32, 4, 51, 51, 258, 450), (8, 4, 32, 32, 384, 512),
    (8, 4, 512, 512, 12, 16), (8, 4, 6, 32, 384, 512), (8, 4, 64, 64, 192, 256), (16, 1, 51, 51, 258, 450)]
if len(sys.argv) > 1 and sys.argv[1] == 'big':
    SHAPES = [(8, 4, 64, 64, 192, 256), (8, 4, 128, 128, 96, 128), (8, 4, 256, 256, 48, 64), (8, 4, 512, 512, 24, 32)]
def timed(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    evs = []
    for _ in range(15):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(1e3 * a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2]
for (N, T, Ci, Co, H, W) in SHAPES:
    x = torch.randn(N, Ci, H, W, device="cuda")
    w = torch.randn(T, Co, Ci, 3, 3, device="cuda") / (3 * Ci ** 0.5)
    b = torch.randn(T, Co, device="cuda")
    fl = 18.0 * Ci * Co * N * H * W
    pf, _ = hip_ops.convk_filters(w, True, False)
    uf, _ = hip_ops.conv3x3_filters(w, True, False)
    yd = hip_ops.convk_tasks_pre(x, pf, T, Ci, Co, 3, b, 0, 0.0, 1)
    yw = hip_ops.conv3x3_tasks_pre(x, uf, T, Ci, Co, b, 0, 0.0, 1)
    d_us = timed(lambda: hip_ops.convk_tasks_pre(x, pf, T, Ci, Co, 3, b, 0, 0.0, 1))
    w_us = timed(lambda: hip_ops.conv3x3_tasks_pre(x, uf, T, Ci, Co, b, 0, 0.0, 1))
    print(json.dumps(dict(layer="%d->%d @%dx%d N=%d T=%d" % (Ci, Co, H, W, N, T), direct_us=round(d_us, 1), direct_TF=round(fl / d_us / 1e6, 1),
                          wino_us=round(w_us, 1), wino_TF=round(fl / w_us / 1e6, 1), rel_diff=float((yd - yw).norm() / yw.norm()))), flush=True)
