import os, sys, time, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from meta_interpolation_amd import synthetic, hip_ops, _hip
from tests.helpers import build_plugin
model = sys.argv[1] if len(sys.argv) > 1 else 'sepconv'
H, W = (256, 448) if model == 'sepconv' else (64, 64)
net = build_plugin(model, 'cuda')
fr = [f.cuda() for f in synthetic.septuplet_batch(2, H, W, model=model)]
names = [n for n, p in net.named_parameters()]
# routed keys: probe with cloned dict
fast = {n: p.detach().clone().requires_grad_() for n, p in net.named_parameters()}
out = net(fr[0], fr[4], params=fast); loss = hip_ops.l1_loss(out, fr[2])
g = torch.autograd.grad(loss, list(fast.values()), allow_unused=True)
routed = [n for n, gi in zip(names, g) if gi is not None]
print(model, 'routed', len(routed), 'of', len(names))
W_in = {n: fast[n].detach().clone().requires_grad_() for n in routed}
lrs = [torch.tensor(1e-3, device='cuda') for _ in routed]
def step():
    out = net(fr[0], fr[4], params=W_in)
    loss = hip_ops.l1_loss(out, fr[2])
    gr = torch.autograd.grad(loss, list(W_in.values()))
    with torch.no_grad():
        new = hip_ops.mt_update(_hip.RULE_SGD, _hip.LR_SCALAR, [w.detach() for w in W_in.values()], list(gr), lrs)
    return loss, gr, new
# eager reference
l0, g0, n0 = step(); torch.cuda.synchronize()
t0 = time.time()
for _ in range(5): step()
torch.cuda.synchronize(); print('eager  %.2f ms/step' % ((time.time() - t0) / 5 * 1e3))
# capture
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
gph = torch.cuda.CUDAGraph()
with torch.cuda.graph(gph):
    l1, g1, n1 = step()
gph.replay(); torch.cuda.synchronize()
print('loss eager %.8f graph %.8f' % (l0.item(), l1.item()))
print('max |dW| ', max((a - b).abs().max().item() for a, b in zip(n0, n1)))
t0 = time.time()
for _ in range(20): gph.replay()
torch.cuda.synchronize(); print('graph  %.2f ms/step' % ((time.time() - t0) / 20 * 1e3))
