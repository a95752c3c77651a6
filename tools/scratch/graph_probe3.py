import os, sys, time, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from meta_interpolation_amd import synthetic, hip_ops, _hip
from tests.helpers import build_plugin
model, mode = sys.argv[1], sys.argv[2]
H, W = (256, 448) if model == 'sepconv' else (64, 64)
net = build_plugin(model, 'cuda')
fr = [f.cuda() for f in synthetic.septuplet_batch(2, H, W, model=model)]
params = [p for p in net.parameters()]
def fn():
    if mode == 'fwd':
        with torch.no_grad():
            return net(fr[0], fr[4]).mean()
    out = net(fr[0], fr[4])
    loss = (out - fr[2]).abs().mean() if mode == 'bwd_torchloss' else hip_ops.l1_loss(out, fr[2])
    gr = torch.autograd.grad(loss, params, allow_unused=True)
    return loss
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): r0 = fn()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
print(model, mode, 'warm ok', flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    r = fn()
print('captured', flush=True)
g.replay(); torch.cuda.synchronize()
print('replay ok', float((r.detach() - r0.detach()).abs().max()), flush=True)
t0 = time.time()
for _ in range(10): g.replay()
torch.cuda.synchronize(); print('graph %.2f ms' % ((time.time() - t0) / 10 * 1e3))
t0 = time.time()
for _ in range(10): fn()
torch.cuda.synchronize(); print('eager %.2f ms' % ((time.time() - t0) / 10 * 1e3))
