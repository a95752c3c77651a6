import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from meta_interpolation_amd import hip_ops
dev = torch.device('cuda')
for (ci, co, h, w) in [(32, 32, 384, 512), (64, 64, 192, 256), (128, 128, 96, 128), (256, 256, 48, 64), (51, 51, 258, 450), (64, 64, 136, 233)]:
    x = torch.randn(2, ci, h, w, device=dev); wt = torch.randn(co, ci, 3, 3, device=dev) / 30; b = torch.randn(co, device=dev)
    for _ in range(5):
        hip_ops.conv3x3(x, wt, b, 0, 0.0)
torch.cuda.synchronize()
