"""The Winograd kernels alone (savfi_conv3x3 forward with task filter sets, savfi_conv3x3_wgrad) on the SepConv layer shapes that stay on
them in config C2, HIP events; SAVFI_HIP_LIB selects a variant library."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meta_interpolation_amd import hip_ops

SHAPES = [(32, 32, 384, 512, 4, 8), (51, 51, 258, 450, 1, 8), (64, 51, 137, 233, 1, 8), (64, 64, 137, 233, 1, 8), (512, 512, 12, 16, 4, 8), (6, 32, 384, 512, 4, 8)]
dev = torch.device("cuda")


def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(1e3 * a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2]


for (ci, co, H, W, T, N) in SHAPES:
    x = torch.randn(N, ci, H, W, device=dev)
    gy = torch.randn(N, co, H, W, device=dev)
    w = torch.randn(T, co, ci, 3, 3, device=dev) / (3 * ci ** 0.5)
    uf, ub = hip_ops.conv3x3_filters(w, True, True)
    fl = 2.0 * 9 * ci * co * H * W * N
    tf = timeit(lambda: hip_ops.conv3x3_tasks_pre(x, uf, T, ci, co, None, 0, 1.0, 1))
    tb = timeit(lambda: hip_ops.conv3x3_tasks_pre(gy, ub, T, ci, co, None, 1, 1.0, 1))
    tw = timeit(lambda: hip_ops.conv3x3_wgrad_tasks(x, gy, T, 1))
    print(json.dumps({"lib": os.path.basename(os.environ.get("SAVFI_HIP_LIB", "default")), "layer": "%d->%d @%dx%d T=%d N=%d" % (ci, co, H, W, T, N),
                      "fwd_us": round(tf, 1), "fwd_TF": round(fl / tf / 1e6, 1), "dgrad_us": round(tb, 1), "dgrad_TF": round(fl / tb / 1e6, 1),
                      "wgrad_us": round(tw, 1), "wgrad_TF": round(fl / tw / 1e6, 1)}), flush=True)
