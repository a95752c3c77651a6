"""A few launches of the direct-convolution kernels on fixed layer shapes, for rocprofv3 --pmc runs (tools/pmc_summary.py reads
the result):   rocprofv3 --pmc <counters> --kernel-trace -d <dir> -- python tools/convk_pmc.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meta_interpolation_amd import hip_ops  # noqa: E402

SHAPES = [(3, 128, 128, 96, 128, 4, 8), (3, 192, 192, 96, 160, 1, 2), (5, 192, 64, 256, 256, 1, 2)]   # K, Ci, Co, H, W, T, N
dev = torch.device("cuda")
for (K, ci, co, H, W, T, N) in SHAPES:
    x = torch.randn(N, ci, H, W, device=dev)
    gy = torch.randn(N, co, H, W, device=dev)
    w = torch.randn(T, co, ci, K, K, device=dev) / (K * ci ** 0.5)
    pf, pb = hip_ops.convk_filters(w, True, True)
    for _ in range(3):
        hip_ops.convk_tasks_pre(x, pf, T, ci, co, K, None, 0, 0.0, K // 2)
        hip_ops.convk_tasks_pre(x, pf, T, ci, co, K, None, 0, 0.0, K // 2, True)
        hip_ops.convk_wgrad_tasks(x, gy, T, K, K // 2)
    if K == 3:
        uf, ub = hip_ops.conv3x3_filters(w, True, False)
        for _ in range(3):
            hip_ops.conv3x3_tasks_pre(x, uf, T, ci, co, None, 0, 0.0, 1)
            hip_ops.conv3x3_wgrad_tasks(x, gy, T, 1)
    torch.cuda.synchronize()
