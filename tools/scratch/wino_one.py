import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from meta_interpolation_amd import hip_ops
dev = torch.device('cuda')
ci, co, h, w = [int(t) for t in (sys.argv[1:5] if len(sys.argv) > 4 else (256, 256, 48, 64))]
x = torch.randn(2, ci, h, w, device=dev); wt = torch.randn(co, ci, 3, 3, device=dev) / 30; b = torch.randn(co, device=dev)
for _ in range(5):
    hip_ops.conv3x3(x, wt, b, 0, 0.0)
torch.cuda.synchronize()
