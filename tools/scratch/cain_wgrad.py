import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from meta_interpolation_amd import hip_ops
from tools.conv_bench import timeit
dev = torch.device('cuda')
for (n, ci, co, h, w, pad) in [(2, 192, 192, 98, 162, 0), (1, 192, 192, 98, 162, 0), (2, 384, 192, 98, 162, 0), (2, 192, 192, 18, 18, 0), (2, 64, 64, 256, 256, 1), (2, 128, 128, 128, 128, 1), (2, 256, 256, 64, 64, 1)]:
    x = torch.randn(n, ci, h, w, device=dev); wt = torch.randn(co, ci, 3, 3, device=dev)
    gy = torch.randn(n, co, h + 2 * pad - 2, w + 2 * pad - 2, device=dev)
    t_mi = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, wt, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False]), 5)
    t_my = timeit(lambda: hip_ops.conv3x3_wgrad(x, gy, pad), 5)
    print((n, ci, co, h, w, pad), 'miopen %.1f us  savfi %.1f us  ratio %.2f' % (t_mi, t_my, t_mi / t_my), flush=True)
