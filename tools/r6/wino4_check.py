"""tools/r6/wino4_check.py -- the Winograd F(4x4, 3x3) kernel (csrc/winograd4.h) against float64 on the GPU, with timings.

    python tools/r6/wino4_check.py [--time]        (SAVFI_HIP_LIB=... for a variant build, e.g. -DSAVFI_W4_MAXC=0 = the F(2x2) kernel)

Every layer shape of at most 64 -> 64 channels: forward (bias, leaky ReLU), data gradient, masked data gradient, per-task filters,
unit-major output (forward) and unit-major cotangent (data gradient).  Prints max / rms error relative to the result's scale.
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from meta_interpolation_amd import hip_ops  # noqa: E402

DEV = 'cuda'


def err(got, want):
    want = want.double()
    d = (got.double().cpu() - want.cpu())
    sc = want.abs().max().item() + 1e-30
    return d.abs().max().item() / sc, d.pow(2).mean().sqrt().item() / sc


def to_unit16(t):
    B, K, H, W = t.shape
    mem = t.reshape(B, K, H, W // 16, 16).permute(0, 2, 3, 1, 4).contiguous()
    return mem.reshape(B, K, H, W)        # same shape, unit-major memory


def from_unit16(t):
    B, K, H, W = t.shape
    return t.reshape(B, H, W // 16, K, 16).permute(0, 3, 1, 2, 4).reshape(B, K, H, W).contiguous()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def check(T, N, Ci, Co, H, W, pad, do_time=False):
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(T, Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)
    b = torch.randn(T, Co, generator=g)
    xc, wc, bc = x.to(DEV), w.to(DEV), b.to(DEV)
    worst = 0.0
    # forward, leaky ReLU
    want = torch.cat([F.leaky_relu(F.conv2d(x[n:n + 1].double(), w[n % T].double(), b[n % T].double(), padding=pad), 0.2) for n in range(N)])
    got = hip_ops.conv3x3_tasks(xc, wc, bc, 0, 0.2, pad)
    e = err(got, want)
    msg = ['fwd %.1e/%.1e' % e]
    worst = max(worst, e[0])
    Ho, Wo = want.shape[2:]
    # data gradient
    gy = torch.randn(N, Co, Ho, Wo, generator=g)
    wantg = torch.cat([F.conv_transpose2d(gy[n:n + 1].double(), w[n % T].double(), padding=pad) for n in range(N)])
    gotg = hip_ops.conv3x3_tasks(gy.to(DEV), wc, None, 1, 1.0, pad)
    e = err(gotg, wantg)
    msg.append('dgrad %.1e/%.1e' % e)
    worst = max(worst, e[0])
    u_f, u_b = hip_ops.conv3x3_filters(wc, True, True)
    assert torch.equal(hip_ops.conv3x3_tasks_pre(xc, u_f, T, Ci, Co, bc, 0, 0.2, pad), got), 'pre fwd != fused'
    assert torch.equal(hip_ops.conv3x3_tasks_pre(gy.to(DEV), u_b, T, Ci, Co, None, 1, 1.0, pad), gotg), 'pre dgrad != fused'
    # masked data gradient
    mask = torch.randn(wantg.shape, generator=g)
    gotm = hip_ops.conv3x3_tasks_pre(gy.to(DEV), u_b, T, Ci, Co, None, 1, 1.0, pad, mask=mask.to(DEV), mask_slope=0.1)
    e = err(gotm, wantg * torch.where(mask > 0, 1.0, 0.1).double())
    msg.append('masked %.1e' % e[0])
    worst = max(worst, e[0])
    # unit-major output / cotangent
    if hip_ops.conv3x3_unit16_supported(xc, wc, pad):
        gu = hip_ops.conv3x3_tasks_pre(xc, u_f, T, Ci, Co, bc, 0, 0.2, pad, out_unit16=True)
        ok = torch.equal(from_unit16(gu), got)
        msg.append('out16 %s' % ('ok' if ok else 'MISMATCH'))
        worst = max(worst, 0.0 if ok else 1.0)
    if Wo % 16 == 0 and hip_ops.conv3x3_in_unit16_supported((N, Co, Ho, Wo), wc, pad):
        gi = hip_ops.conv3x3_dgrad_in_unit16(to_unit16(gy.to(DEV)), u_b, T, Ci, Co, pad)
        ok = torch.equal(gi, gotg)
        msg.append('in16 %s' % ('ok' if ok else 'MISMATCH %.1e' % err(gi, wantg)[0]))
        worst = max(worst, 0.0 if ok else 1.0)
    line = 'T%d N%d %d->%d %dx%d pad%d: %s' % (T, N, Ci, Co, H, W, pad, '  '.join(msg))
    if do_time:
        gyc = gy.to(DEV)
        tf = timeit(lambda: hip_ops.conv3x3_tasks_pre(xc, u_f, T, Ci, Co, bc, 0, 0.2, pad))
        tb = timeit(lambda: hip_ops.conv3x3_tasks_pre(gyc, u_b, T, Ci, Co, None, 1, 1.0, pad))
        fl = 18.0 * Ci * Co * Ho * Wo * N
        line += '   fwd %.0f us (%.0f TF)  dgrad %.0f us (%.0f TF)' % (tf, fl / tf / 1e6, tb, 18.0 * Ci * Co * H * W * N / tb / 1e6 if pad == 1 else fl / tb / 1e6)
    print(line, flush=True)
    return worst


SMALL = [
    (1, 1, 3, 5, 5, 7, 1), (1, 1, 3, 5, 5, 7, 0), (1, 2, 6, 32, 24, 40, 1), (1, 2, 6, 32, 24, 40, 0), (1, 1, 8, 64, 16, 64, 1),
    (1, 1, 64, 51, 37, 45, 1), (1, 1, 64, 51, 37, 45, 0), (4, 8, 51, 51, 18, 30, 1), (2, 4, 51, 51, 18, 34, 0), (2, 4, 64, 32, 16, 64, 0),
    (1, 2, 32, 32, 16, 16, 1), (4, 8, 32, 32, 20, 30, 1), (2, 2, 40, 64, 9, 130, 1), (1, 1, 33, 17, 130, 9, 1), (1, 1, 16, 16, 4, 4, 1),
    (1, 1, 16, 16, 6, 6, 0), (4, 4, 51, 51, 34, 66, 0), (2, 2, 51, 51, 32, 64, 1), (1, 1, 51, 51, 66, 130, 0),
    # deep layers on small maps: the reduction split over workgroups (partial outputs + wino_split_reduce)
    (4, 8, 256, 256, 24, 32, 1), (2, 4, 512, 512, 12, 16, 1), (1, 2, 512, 256, 24, 32, 0), (4, 8, 320, 192, 13, 21, 1),
]
BIG = [(4, 8, 32, 32, 384, 512, 1), (4, 32, 64, 51, 137, 236, 1), (4, 32, 51, 51, 258, 450, 0), (4, 8, 32, 64, 192, 256, 1),
       (4, 16, 51, 51, 258, 450, 0), (4, 8, 64, 64, 192, 256, 1)]

if __name__ == '__main__':
    do_time = '--time' in sys.argv
    worst = 0.0
    for c in SMALL:
        worst = max(worst, check(*c))
    if do_time:
        for c in BIG:
            worst = max(worst, check(*c, do_time=True))
    print('WORST', worst)
