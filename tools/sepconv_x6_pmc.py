"""PMC driver: the sepconv filter-gradient kernel at B=8 256x448 (the bench shape), a few launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meta_interpolation_amd import _hip
lib, st = _hip.lib(), _hip.current_stream()
B, C, Ho, Wo, K = 8, 3, 256, 448, 51
F8 = len(sys.argv) > 1 and sys.argv[1] == "f8"        # frames of 8-bit images through the frames8 entry points (the three-product kernels)
inp = torch.randint(0, 256, (B, C, Ho + K - 1, Wo + K - 1), device="cuda").float().div(255) if F8 else torch.rand(B, C, Ho + K - 1, Wo + K - 1, device="cuda")
words = torch.empty(256, dtype=torch.int32, device="cuda")
lib.savfi_frames8_classify_f32(inp.data_ptr(), inp.numel(), words.data_ptr(), st)
v = torch.randn(B, K, Ho, Wo, device="cuda") / 7
h = torch.randn(B, K, Ho, Wo, device="cuda") / 7
gO = torch.randn(B, C, Ho, Wo, device="cuda")
gV, gH = torch.empty_like(v), torch.empty_like(h)
for _ in range(6):
    if F8:
        _hip.check(lib.savfi_sepconv_bwd_frames8_f32(inp.data_ptr(), v.data_ptr(), h.data_ptr(), gO.data_ptr(), gV.data_ptr(), gH.data_ptr(), words.data_ptr(), B, C, Ho, Wo, K, K, 0, st), "bwd8")
        continue
    _hip.check(lib.savfi_sepconv_bwd_f32(inp.data_ptr(), v.data_ptr(), h.data_ptr(), gO.data_ptr(), None, gV.data_ptr(), gH.data_ptr(), B, C, Ho, Wo, K, st), "bwd")
out = torch.empty_like(gO)
for _ in range(6):
    if F8:
        _hip.check(lib.savfi_sepconv_fwd_frames8_f32(inp.data_ptr(), v.data_ptr(), h.data_ptr(), out.data_ptr(), words.data_ptr(), B, C, Ho, Wo, K, K, 0, st), "fwd8")
        continue
    _hip.check(lib.savfi_sepconv_fwd_f32(inp.data_ptr(), v.data_ptr(), h.data_ptr(), out.data_ptr(), B, C, Ho, Wo, K, st), "fwd")
torch.cuda.synchronize()
