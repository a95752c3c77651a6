import os, sys, random, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from meta_interpolation_amd import hip_ops
dev = torch.device('cuda')
random.seed(1); torch.manual_seed(1)
bad = 0
for it in range(300):
    n = random.choice([1, 1, 2, 3])
    ci = random.choice([3, 6, 8, 16, 32, 51, 64, 96, 128, 192, 256])
    co = random.choice([3, 8, 16, 32, 51, 64, 96, 128, 192, 256])
    h = random.choice([5, 16, 31, 64, 96, 128, 130, 192, 256, 258])
    w = random.choice([4, 7, 16, 33, 64, 100, 128, 233, 256, 450, 512])
    pad = random.choice([0, 1, 1])
    mode = random.choice([0, 1])
    if pad == 0 and (h < 3 or w < 3):
        continue
    if mode == 0:
        x = torch.randn(n, ci, h, w, device=dev)
    else:
        x = torch.randn(n, co, h, w, device=dev)
    wt = torch.randn(co, ci, 3, 3, device=dev) / (3 * (ci if mode == 0 else co) ** 0.5)
    b = torch.randn(co, device=dev) if mode == 0 else None
    junk = torch.full((1 << 20,), float('nan'), device=dev); del junk
    o1 = hip_ops.conv3x3(x, wt, b, mode, 1.0, pad)
    junk = torch.full((1 << 21,), float('nan'), device=dev); del junk
    o2 = hip_ops.conv3x3(x, wt, b, mode, 1.0, pad)
    if mode == 0:
        want = F.conv2d(x.double(), wt.double(), b.double(), padding=pad)
    else:
        xin = torch.zeros(n, ci, h + 2 - 2 * pad, w + 2 - 2 * pad, device=dev, dtype=torch.double, requires_grad=True)
        (want,) = torch.autograd.grad(F.conv2d(xin, wt.double(), None, padding=pad), xin, x.double())
    err = float((o1.double() - want).abs().max() / want.abs().max().clamp_min(1e-30))
    same = torch.equal(o1, o2)
    if err > 3e-6 or not same or torch.isnan(o1).any():
        bad += 1
        print("BAD", (n, ci, co, h, w, pad, mode), "err", err, "same", same, flush=True)
print("done, bad =", bad)
