import os, sys, time, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from meta_interpolation_amd import hip_ops, _hip
which = sys.argv[1]
x = torch.randn(2, 16, 64, 64, device='cuda', requires_grad=True)
w = torch.randn(16, 16, 3, 3, device='cuda', requires_grad=True)
t = torch.randn(2, 16, 64, 64, device='cuda')
def f_conv():
    y = torch.nn.functional.conv2d(x, w, None, 1, 1)
    return (y - t).abs().mean()
def f_conv_bwd():
    l = f_conv(); return torch.autograd.grad(l, [w])[0]
def f_custom():
    return hip_ops.l1_loss(x, t)
def f_custom_bwd():
    l = hip_ops.l1_loss(torch.nn.functional.conv2d(x, w, None, 1, 1), t); return torch.autograd.grad(l, [w])[0]
LR = torch.tensor(1e-3, device='cuda')
def f_update():
    with torch.no_grad():
        return hip_ops.mt_update(_hip.RULE_SGD, _hip.LR_SCALAR, [w.detach()], [w.detach()], [LR])[0]
def f_shuffle():
    return hip_ops.pixel_shuffle(hip_ops.pixel_shuffle(x[:, :3].contiguous(), 1 / 8), 8).sum()
def f_mean():
    return hip_ops.mt_mean([w.detach(), x.detach()])
fn = {'conv': f_conv, 'conv_bwd': f_conv_bwd, 'custom': f_custom, 'custom_bwd': f_custom_bwd, 'update': f_update, 'shuffle': f_shuffle, 'mean': f_mean}[which]
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): r0 = fn()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
print(which, 'warm ok', flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    r = fn()
print(which, 'captured', flush=True)
g.replay(); torch.cuda.synchronize()
print(which, 'replay ok', float((r - r0).abs().max()), flush=True)
