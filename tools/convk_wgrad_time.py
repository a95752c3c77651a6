"""HIP-event timing of savfi_convk_wgrad_tasks_reflect_f32 (direct split-bf16 weight gradient) on a few layer shapes.
python tools/convk_wgrad_time.py [set]   (sets: cain = CAIN's 192->192 @96x160 N=2 with the mirrored border; sepconv)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meta_interpolation_amd import hip_ops
SETS = {"cain": [(2, 192, 192, 96, 160, 3, 1, True), (1, 192, 192, 96, 160, 3, 1, True), (2, 192, 192, 16, 16, 3, 1, False)],
        "sepconv": [(8, 32, 32, 384, 512, 3, 1, False), (8, 64, 64, 192, 256, 3, 1, False), (8, 256, 256, 48, 64, 3, 1, False)]}
for (N, Ci, Co, H, W, K, pad, reflect) in SETS[sys.argv[1] if len(sys.argv) > 1 else "cain"]:
    x = torch.randn(N, Ci, H, W, device="cuda")
    gz = torch.randn(N, Co, H + 2 * pad - K + 1, W + 2 * pad - K + 1, device="cuda")
    f = lambda: hip_ops.convk_wgrad_tasks(x, gz, 1, K, pad, False, reflect)
    for _ in range(3): f()
    torch.cuda.synchronize()
    evs = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(1e3 * a.elapsed_time(b) for a, b in evs)
    fl = 2.0 * K * K * Ci * Co * N * gz.shape[2] * gz.shape[3]
    print(json.dumps(dict(layer="%dx%d %d->%d @%dx%d N=%d%s" % (K, K, Ci, Co, H, W, N, " reflect" if reflect else ""), median_us=round(t[len(t) // 2], 1),
                          TFLOPs=round(fl / t[len(t) // 2] / 1e6, 1), target=os.environ.get("SAVFI_WGRAD_TARGET"), ng=os.environ.get("SAVFI_WGRAD_NG"))), flush=True)
