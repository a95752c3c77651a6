"""HIP-event timing of the 3x3 weight gradient as the plugin calls it (hip_ops.conv3x3_wgrad_tasks: Winograd form where eligible) on the
task-batched SepConv layer shapes (T = 4 filter sets, N = 8), small maps included.  Env: SAVFI_WWGRAD_MIN_CHUNKS."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meta_interpolation_amd import hip_ops
LAYERS = [(8, 4, 32, 384, 512), (8, 4, 64, 192, 256), (8, 4, 128, 96, 128), (8, 4, 256, 48, 64), (8, 4, 512, 24, 32), (8, 4, 512, 12, 16),
          (8, 4, 256, 24, 32), (8, 4, 64, 96, 128), (8, 4, 128, 48, 64)]
for (N, T, C, H, W) in LAYERS:
    x = torch.randn(N, C, H, W, device="cuda")
    gz = torch.randn(N, C, H, W, device="cuda")
    f = lambda: hip_ops.conv3x3_wgrad_tasks(x, gz, T, 1)
    for _ in range(3): f()
    torch.cuda.synchronize()
    evs = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(1e3 * a.elapsed_time(b) for a, b in evs)
    fl = 18.0 * C * C * N * H * W
    print(json.dumps(dict(layer="%d->%d @%dx%d T=%d N=%d" % (C, C, H, W, T, N), median_us=round(t[len(t) // 2], 1), TFLOPs=round(fl / t[len(t) // 2] / 1e6, 1),
                          min_chunks=os.environ.get("SAVFI_WWGRAD_MIN_CHUNKS"))), flush=True)
