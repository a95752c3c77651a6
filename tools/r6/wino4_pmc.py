"""A few launches of the Winograd F(4x4, 3x3) kernel on the three biggest layer shapes of config C2, for rocprofv3 --pmc runs
(tools/pmc_summary.py reads the result; kernel name filter "wino4")."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from meta_interpolation_amd import hip_ops  # noqa: E402

for (N, T, ci, co, H, W, pad) in [(32, 4, 51, 51, 258, 450, 0), (8, 4, 64, 64, 192, 256, 1), (8, 4, 128, 128, 96, 128, 1)]:
    x = torch.randn(N, ci, H, W, device="cuda")
    w = torch.randn(T, co, ci, 3, 3, device="cuda") / (3 * ci ** 0.5)
    b = torch.randn(T, co, device="cuda")
    u_f, u_b = hip_ops.conv3x3_filters(w, True, True)
    gy = torch.randn(N, co, H + 2 * pad - 2, W + 2 * pad - 2, device="cuda")
    for _ in range(3):
        hip_ops.conv3x3_tasks_pre(x, u_f, T, ci, co, b, 0, 0.0, pad)
        hip_ops.conv3x3_tasks_pre(gy, u_b, T, ci, co, None, 1, 1.0, pad)
    torch.cuda.synchronize()
