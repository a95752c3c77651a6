import os, sys, torch
import torch.nn.functional as F
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from meta_interpolation_amd import hip_ops
from meta_interpolation_amd.sepconv.sepconv_op.sepconv import FunctionSepconv
dev = torch.device("cuda"); torch.manual_seed(0)
if os.environ.get("DET"): torch.backends.cudnn.deterministic = True
REPS = 60
def check(name, fn):
    ref = fn()
    ref = [r.clone() for r in (ref if isinstance(ref, (tuple, list)) else [ref])]
    bad = 0; worst = 0.0
    for _ in range(REPS):
        out = fn()
        out = out if isinstance(out, (tuple, list)) else [out]
        for a, b in zip(out, ref):
            if not torch.equal(a, b):
                bad += 1; worst = max(worst, float((a - b).abs().max() / (b.abs().max() + 1e-30))); break
    print('%-46s mismatching runs %3d / %d   worst rel %.2e' % (name, bad, REPS, worst), flush=True)

for (B, Ho, Wo) in [(2, 128, 128), (1, 128, 128), (2, 78, 60), (2, 64, 64), (1, 256, 448), (2, 256, 448), (2, 384, 512)]:
    K = 51
    inp = torch.rand(B, 3, Ho + K - 1, Wo + K - 1, device=dev)
    v = (torch.randn(B, K, Ho, Wo, device=dev) / 7).requires_grad_(); h = (torch.randn(B, K, Ho, Wo, device=dev) / 7).requires_grad_()
    gO = torch.randn(B, 3, Ho, Wo, device=dev)
    check('sepconv fwd B=%d %dx%d' % (B, Ho, Wo), lambda: FunctionSepconv.apply(inp, v, h).detach())
    check('sepconv bwd B=%d %dx%d' % (B, Ho, Wo), lambda: torch.autograd.grad(FunctionSepconv.apply(inp, v, h), [v, h], gO))
x = torch.randn(2, 51, 64, 64, device=dev, requires_grad=True)
check('upsample2x fwd', lambda: hip_ops.upsample_bilinear2x(x, True).detach())
go = torch.randn(2, 51, 128, 128, device=dev)
check('upsample2x bwd', lambda: torch.autograd.grad(hip_ops.upsample_bilinear2x(x, True), x, go))
for (ci, co, hh, ww) in [(6, 32, 128, 128), (32, 32, 128, 128), (64, 64, 64, 64), (64, 51, 64, 64), (51, 51, 128, 128), (512, 512, 4, 4), (256, 256, 16, 16)]:
    xx = torch.randn(2, ci, hh, ww, device=dev, requires_grad=True); wt = (torch.randn(co, ci, 3, 3, device=dev) / 10).requires_grad_(); b = torch.randn(co, device=dev, requires_grad=True)
    gy = torch.randn(2, co, hh, ww, device=dev)
    check('miopen conv fwd %d->%d @%dx%d' % (ci, co, hh, ww), lambda: F.conv2d(xx, wt, b, padding=1).detach())
    check('miopen conv dgrad %d->%d @%dx%d' % (ci, co, hh, ww), lambda: torch.autograd.grad(F.conv2d(xx, wt, b, padding=1), xx, gy))
    check('miopen conv wgrad+bias %d->%d @%dx%d' % (ci, co, hh, ww), lambda: torch.autograd.grad(F.conv2d(xx, wt, b, padding=1), [wt, b], gy))
xx = torch.randn(2, 3, 64, 64, device=dev, requires_grad=True)
check('replicate pad bwd', lambda: torch.autograd.grad(F.pad(xx, (25, 39, 25, 39), mode='replicate'), xx, torch.randn(2, 3, 128, 128, device=dev).fill_(0.37)))
xx = torch.randn(2, 32, 128, 128, device=dev, requires_grad=True)
gp = torch.randn(2, 32, 64, 64, device=dev)
check('avg_pool bwd', lambda: torch.autograd.grad(F.avg_pool2d(xx, 2, 2), xx, gp))
