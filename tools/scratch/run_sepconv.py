import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from meta_interpolation_amd import _hip
which = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B, C, Ho, Wo, K = 1, 3, 384, 512, 51
lib, st = _hip.lib(), _hip.current_stream()
inp = torch.rand(B, C, Ho + K - 1, Wo + K - 1, device='cuda'); v = torch.randn(B, K, Ho, Wo, device='cuda') / 7
h = torch.randn(B, K, Ho, Wo, device='cuda') / 7; gO = torch.randn(B, C, Ho, Wo, device='cuda')
out, gV, gH = torch.empty_like(gO), torch.empty_like(v), torch.empty_like(h)
for _ in range(n):
    if which == 'fwd':
        lib.savfi_sepconv_fwd_f32(inp.data_ptr(), v.data_ptr(), h.data_ptr(), out.data_ptr(), B, C, Ho, Wo, K, st)
    else:
        lib.savfi_sepconv_bwd_f32(inp.data_ptr(), v.data_ptr(), h.data_ptr(), gO.data_ptr(), None, gV.data_ptr(), gH.data_ptr(), B, C, Ho, Wo, K, st)
torch.cuda.synchronize()
