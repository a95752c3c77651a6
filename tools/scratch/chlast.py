import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from meta_interpolation_amd import synthetic
from tests.helpers import build_plugin
net = build_plugin('sepconv', 'cuda')
fr = [f.cuda() for f in synthetic.septuplet_batch(2, 256, 448)]
def run(n=5):
    for it in range(n + 2):
        if it == 2:
            torch.cuda.synchronize(); t0 = time.time()
        out = net(fr[0], fr[4])
        loss = (out - fr[2]).abs().mean()
        g = torch.autograd.grad(loss, [p for p in net.parameters()], allow_unused=True)
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3
print('NCHW      fwd+bwd N=2: %.1f ms' % run())
net = net.to(memory_format=torch.channels_last)
fr = [f.contiguous(memory_format=torch.channels_last) for f in fr]
print('NHWC      fwd+bwd N=2: %.1f ms' % run())
