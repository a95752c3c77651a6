import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from meta_interpolation_amd import hip_ops
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for T, N, Ci, Co, H, W, pad in [(4, 8, 64, 64, 192, 256, 1), (4, 8, 128, 128, 96, 128, 1), (4, 8, 256, 256, 48, 64, 1), (4, 8, 32, 32, 384, 512, 1), (1, 2, 192, 192, 96, 160, 0)]:
    w = torch.randn(T, Co, Ci, 3, 3, device='cuda') / (3 * Ci ** 0.5)
    u_f, u_b = hip_ops.conv3x3_filters(w, True, True)
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    gy = torch.randn(N, Co, Ho, Wo, device='cuda')
    mask = torch.randn(N, Ci, H, W, device='cuda')
    t0 = timeit(lambda: hip_ops.conv3x3_tasks_pre(gy, u_b, T, Ci, Co, None, 1, 1.0, pad))
    t1 = timeit(lambda: hip_ops.conv3x3_tasks_pre(gy, u_b, T, Ci, Co, None, 1, 1.0, pad, mask=mask, mask_slope=0.0))
    print("%d->%d @%dx%d N%d pad %d: data gradient %.1f us | masked %.1f us" % (Ci, Co, H, W, N, pad, t0, t1))
