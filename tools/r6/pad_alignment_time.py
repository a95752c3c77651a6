"""pad 1 on the map against pad 0 on a padded copy of it, and the share of border workgroups (same tile count, wider maps), F(4x4) forward."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from meta_interpolation_amd import hip_ops  # noqa: E402


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for T, N, Ci, Co, H, W in [(4, 8, 64, 64, 192, 256), (4, 8, 64, 64, 96, 512), (4, 8, 64, 64, 48, 1024), (4, 8, 64, 64, 24, 2048),
                           (4, 8, 128, 128, 96, 128), (1, 2, 192, 192, 96, 160), (4, 8, 256, 256, 48, 64)]:
    w = torch.randn(T, Co, Ci, 3, 3, device='cuda') / (3 * Ci ** 0.5)
    b = torch.randn(T, Co, device='cuda')
    u_f, u_b = hip_ops.conv3x3_filters(w, True, True)
    x1 = torch.randn(N, Ci, H, W, device='cuda')
    x0 = torch.randn(N, Ci, H + 2, W + 2, device='cuda')
    t1 = timeit(lambda: hip_ops.conv3x3_tasks_pre(x1, u_f, T, Ci, Co, b, 0, 0.0, 1))
    t0 = timeit(lambda: hip_ops.conv3x3_tasks_pre(x0, u_f, T, Ci, Co, b, 0, 0.0, 0))
    print("%d->%d @%dx%d N%d: pad 1 on the map %.1f us | pad 0 on a padded copy %.1f us" % (Ci, Co, H, W, N, t1, t0))
