"""HIP-event timing of the last Subnet convolution's data gradient (51 -> 51 @256x448 -> 258x450, T = 4, N = 32): planar cotangent
(savfi_conv3x3_tasks_pre_f32 mode 1) against the unit-major one (savfi_conv3x3_dgrad_in_unit16_f32)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meta_interpolation_amd import hip_ops
N, T, C, H, W = 32, 4, 51, 256, 448
gy = torch.randn(N, C, H, W, device="cuda")
w = torch.randn(T, C, C, 3, 3, device="cuda") / 21
u = hip_ops.conv3x3_filters(w, False, True)[1]
runs = {"planar": lambda: hip_ops.conv3x3_tasks_pre(gy, u, T, C, C, None, 1, 1.0, 0), "unit16": lambda: hip_ops.conv3x3_dgrad_in_unit16(gy, u, T, C, C, 0)}
for name, f in runs.items():
    for _ in range(3): f()
    torch.cuda.synchronize()
    evs = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(1e3 * a.elapsed_time(b) for a, b in evs)
    print(json.dumps(dict(layout=name, median_us=round(t[len(t) // 2], 1), TFLOPs=round(18.0 * C * C * H * W * N / t[len(t) // 2] / 1e6, 1))), flush=True)
