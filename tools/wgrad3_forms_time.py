"""HIP-event timing of the 3 x 3 weight gradient in its forms on the layer shapes of the bench configs:
Winograd F(3x3, 2x2) on the fp32 matrix cores (savfi_conv3x3_wgrad_wino_tasks_f32) against the direct split-bf16 kernel
(savfi_convk_wgrad_tasks_f32; SAVFI_WGRAD3_ALLTAPS=0 in the environment selects its tap-split form).
python tools/wgrad3_forms_time.py [c2|c5]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meta_interpolation_amd import hip_ops
SETS = {  # N, T, Ci, Co, H, W
    "c2": [(8, 4, 64, 64, 192, 256), (8, 4, 128, 128, 96, 128), (8, 4, 256, 256, 48, 64), (8, 4, 512, 512, 24, 32), (8, 4, 512, 512, 12, 16),
           (8, 4, 64, 64, 96, 128), (8, 4, 128, 128, 48, 64), (8, 4, 256, 256, 24, 32), (8, 4, 128, 64, 96, 128), (8, 4, 64, 128, 96, 128),
           (8, 4, 512, 256, 24, 32), (8, 4, 256, 512, 24, 32), (8, 4, 256, 128, 48, 64), (8, 4, 128, 256, 48, 64),
           (32, 4, 64, 64, 137, 236), (32, 4, 64, 51, 137, 236), (32, 4, 51, 51, 258, 450), (16, 1, 51, 51, 258, 450), (8, 4, 32, 32, 384, 512)],
    "c2s": [(8, 4, 128, 128, 96, 128), (8, 4, 64, 64, 192, 256), (32, 4, 64, 64, 137, 236)],
    "c5": [(2, 1, 192, 192, 96, 160), (1, 1, 192, 192, 96, 160), (2, 1, 192, 192, 16, 16)],
}
def timed(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    evs = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(1e3 * a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2]
for (N, T, Ci, Co, H, W) in SETS[sys.argv[1] if len(sys.argv) > 1 else "c2"]:
    x = torch.randn(N, Ci, H, W, device="cuda")
    gz = torch.randn(N, Co, H, W, device="cuda")
    fl = 18.0 * Ci * Co * N * H * W
    row = dict(layer="%d->%d @%dx%d N=%d T=%d" % (Ci, Co, H, W, N, T), alltaps=os.environ.get("SAVFI_WGRAD3_ALLTAPS", "1"))
    d = hip_ops.convk_wgrad_tasks(x, gz, T, 3, 1)
    us = timed(lambda: hip_ops.convk_wgrad_tasks(x, gz, T, 3, 1))
    row.update(direct_us=round(us, 1), direct_TF=round(fl / us / 1e6, 1))
    if hip_ops._wgrad_wino(N, Ci, Co, H, W):
        wv = hip_ops.conv3x3_wgrad_tasks(x, gz, T, 1)
        us = timed(lambda: hip_ops.conv3x3_wgrad_tasks(x, gz, T, 1))
        row.update(wino_us=round(us, 1), wino_TF=round(fl / us / 1e6, 1), rel_diff=float((d - wv).norm() / wv.norm()))
    print(json.dumps(row), flush=True)
