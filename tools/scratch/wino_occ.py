import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from meta_interpolation_amd import hip_ops
dev = torch.device('cuda')
def t(n, ci=64, co=32, h=64, w=512):
    x = torch.randn(n, ci, h, w, device=dev); wt = torch.randn(co, ci, 3, 3, device=dev) / 30; b = torch.randn(co, device=dev)
    for _ in range(3): hip_ops.conv3x3(x, wt, b, 0, 0.0)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, bb in ev:
        a.record(); hip_ops.conv3x3(x, wt, b, 0, 0.0); bb.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(bb) * 1e3 for a, bb in ev)[5]
for n in (1, 2, 3, 4, 6, 8):
    print("WGs", n * 128, "time us", round(t(n), 1), flush=True)
