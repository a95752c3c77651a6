"""A few launches of the 3 x 3 weight-gradient forms on two layer shapes, for rocprofv3 --pmc runs (tools/pmc_summary.py reads the result)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meta_interpolation_amd import hip_ops  # noqa: E402
for (N, T, ci, co, H, W) in [(8, 4, 128, 128, 96, 128), (2, 1, 192, 192, 96, 160)]:
    x = torch.randn(N, ci, H, W, device="cuda")
    gy = torch.randn(N, co, H, W, device="cuda")
    for _ in range(3):
        hip_ops.convk_wgrad_tasks(x, gy, T, 3, 1)
        hip_ops.conv3x3_wgrad_tasks(x, gy, T, 1)
    torch.cuda.synchronize()
