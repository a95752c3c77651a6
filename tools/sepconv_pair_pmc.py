"""PMC driver of the shipped SepConv kernels (round 5): the pair launches of the plugin's tail at B = 8, 256 x 448 -- 16 virtual samples,
frames of 8-bit images, taps and (backward) gradients unit-major: sepconv_bwd_ws<U8, DMA> / sepconv_fwd_ws<U8> plus the early exits of the
six-product instances.  Six launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meta_interpolation_amd import _hip
lib, st = _hip.lib(), _hip.current_stream()
B, C, Ho, Wo, K = 8, 3, 256, 448, 51
fr = [torch.randint(0, 256, (B, C, Ho + K - 1, Wo + K - 1), device="cuda").float().div(255) for _ in range(2)]
words = [torch.empty(256, dtype=torch.int32, device="cuda") for _ in range(2)]
for f, w in zip(fr, words):
    lib.savfi_frames8_classify_f32(f.data_ptr(), f.numel(), w.data_ptr(), st)
taps = torch.randn(4 * B, K, Ho, Wo, device="cuda") / 7
gO = torch.randn(B, C, Ho, Wo, device="cuda")
gT, out2 = torch.empty_like(taps), torch.empty(B, 2, C, Ho, Wo, device="cuda")
for _ in range(6):
    _hip.check(lib.savfi_sepconv_bwd_pair_frames8_f32(fr[0].data_ptr(), fr[1].data_ptr(), taps.data_ptr(), gO.data_ptr(), gT.data_ptr(),
                                                      words[0].data_ptr(), words[1].data_ptr(), B, C, Ho, Wo, K, 3, st), "bwd pair")
for _ in range(6):
    _hip.check(lib.savfi_sepconv_fwd_pair_frames8_f32(fr[0].data_ptr(), fr[1].data_ptr(), taps.data_ptr(), out2.data_ptr(), words[0].data_ptr(),
                                                      words[1].data_ptr(), B, C, Ho, Wo, K, 1, st), "fwd pair")
torch.cuda.synchronize()
