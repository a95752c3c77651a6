import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from meta_interpolation_amd import _hip
B, C, Ho, Wo, K = 1, 3, 384, 512, 51
lib, st = _hip.lib(), _hip.current_stream()
inp = torch.rand(B, C, Ho + K - 1, Wo + K - 1, device='cuda'); v = torch.randn(B, K, Ho, Wo, device='cuda') / 7
h = torch.randn(B, K, Ho, Wo, device='cuda') / 7; gO = torch.randn(B, C, Ho, Wo, device='cuda')
gV, gH = torch.empty_like(v), torch.empty_like(h)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
P = lambda x: None if x is None else x.data_ptr()
call = lambda gv, gh: lib.savfi_sepconv_bwd_f32(inp.data_ptr(), v.data_ptr(), h.data_ptr(), gO.data_ptr(), None, P(gv), P(gh), B, C, Ho, Wo, K, st)
print('both  %.1f us' % t(lambda: call(gV, gH)))
print('gV    %.1f us' % t(lambda: call(gV, None)))
print('gH    %.1f us' % t(lambda: call(None, gH)))
