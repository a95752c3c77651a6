"""tools/r6/wino4_time.py -- timings only (for the -DW4_ABL=n ablation builds of csrc/winograd4.h, whose results are wrong)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from meta_interpolation_amd import hip_ops  # noqa: E402

SHAPES = [(4, 8, 32, 32, 384, 512, 1), (4, 32, 64, 51, 137, 236, 1), (4, 32, 51, 51, 258, 450, 0), (4, 8, 32, 64, 192, 256, 1), (4, 8, 64, 64, 192, 256, 1)]
if len(sys.argv) > 1 and sys.argv[1] == 'small':
    SHAPES = [(4, 8, 512, 512, 24, 32, 1), (1, 8, 512, 512, 24, 32, 1), (4, 8, 256, 256, 24, 32, 1), (4, 8, 512, 256, 24, 32, 1), (4, 8, 256, 512, 24, 32, 1),
              (4, 8, 512, 512, 12, 16, 1), (4, 8, 128, 128, 48, 64, 1), (4, 8, 64, 64, 96, 128, 1)]
elif len(sys.argv) > 1 and sys.argv[1] == 'deep':
    SHAPES = [(4, 8, 64, 64, 192, 256, 1), (4, 8, 128, 128, 96, 128, 1), (4, 8, 256, 256, 48, 64, 1), (4, 32, 64, 64, 137, 236, 1), (4, 8, 64, 64, 96, 128, 1),
              (4, 8, 128, 128, 48, 64, 1), (4, 8, 128, 64, 96, 128, 1), (4, 8, 64, 128, 96, 128, 1)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


out = []
for T, N, Ci, Co, H, W, pad in SHAPES:
    x = torch.randn(N, Ci, H, W, device='cuda')
    w = torch.randn(T, Co, Ci, 3, 3, device='cuda') / (3 * Ci ** 0.5)
    b = torch.randn(T, Co, device='cuda')
    u_f, u_b = hip_ops.conv3x3_filters(w, True, True)
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    gy = torch.randn(N, Co, Ho, Wo, device='cuda')
    tf = timeit(lambda: hip_ops.conv3x3_tasks_pre(x, u_f, T, Ci, Co, b, 0, 0.2, pad))
    tb = timeit(lambda: hip_ops.conv3x3_tasks_pre(gy, u_b, T, Ci, Co, None, 1, 1.0, pad))
    fl = 18.0 * Ci * Co * Ho * Wo * N
    line = '%d->%d@%dx%d N%d: %.0f / %.0f us (%.0f / %.0f TF)' % (Ci, Co, H, W, N, tf, tb, fl / tf / 1e6, fl / tb / 1e6)
    if 'convk' in sys.argv:          # the direct split-bf16 kernel on the same box
        p_f, p_b = hip_ops.convk_filters(w, True, True)
        kf = timeit(lambda: hip_ops.convk_tasks_pre(x, p_f, T, Ci, Co, 3, b, 0, 0.2, pad))
        kb = timeit(lambda: hip_ops.convk_tasks_pre(gy, p_b, T, Ci, Co, 3, None, 1, 1.0, pad))
        line += ' [convk %.0f / %.0f us]' % (kf, kb)
    out.append(line)
print(os.environ.get('SAVFI_HIP_LIB', 'default'), ' | '.join(out), flush=True)
