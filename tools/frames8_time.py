"""HIP-event timing of the SepConv op at B x 256 x 448 on frames of 8-bit images: savfi_sepconv_{fwd,bwd}_frames8_f32 with the frame's own
words (three-product kernels) against the entry points without the words (six-product kernels); isolated launches, random taps."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meta_interpolation_amd import _hip
from meta_interpolation_amd.sepconv.sepconv_op import sepconv as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
C, K = 3, 51
Ho = int(sys.argv[2]) if len(sys.argv) > 2 else 256
Wo = int(sys.argv[3]) if len(sys.argv) > 3 else 448
ONLY = sys.argv[4].split(",") if len(sys.argv) > 4 else None
lib, st = _hip.lib(), _hip.current_stream()
inp = torch.randint(0, 256, (B, C, Ho + K - 1, Wo + K - 1), device="cuda").float().div(255)
v = torch.randn(B, K, Ho, Wo, device="cuda") / 7
h = torch.randn(B, K, Ho, Wo, device="cuda") / 7
gO = torch.randn(B, C, Ho, Wo, device="cuda")
gV, gH, out = torch.empty_like(v), torch.empty_like(h), torch.empty_like(gO)
words = S.frames8_classify(inp)
P = lambda t: t.data_ptr()
runs = {
    "bwd_six": lambda: lib.savfi_sepconv_bwd_taps_strided_f32(P(inp), P(v), P(h), P(gO), P(gV), P(gH), B, C, Ho, Wo, K, K, st),
    "bwd_frames8": lambda: lib.savfi_sepconv_bwd_frames8_f32(P(inp), P(v), P(h), P(gO), P(gV), P(gH), P(words), B, C, Ho, Wo, K, K, 0, st),
    "fwd_six": lambda: lib.savfi_sepconv_fwd_taps_strided_f32(P(inp), P(v), P(h), P(out), B, C, Ho, Wo, K, K, st),
    "fwd_frames8": lambda: lib.savfi_sepconv_fwd_frames8_f32(P(inp), P(v), P(h), P(out), P(words), B, C, Ho, Wo, K, K, 0, st),
    "bwd_frames8_unit16": lambda: lib.savfi_sepconv_bwd_frames8_f32(P(inp), P(v), P(h), P(gO), P(gV), P(gH), P(words), B, C, Ho, Wo, K, K, 3, st),
    "fwd_frames8_unit16": lambda: lib.savfi_sepconv_fwd_frames8_f32(P(inp), P(v), P(h), P(out), P(words), B, C, Ho, Wo, K, K, 1, st),
    "classify": lambda: lib.savfi_frames8_classify_f32(P(inp), inp.numel(), P(words), st),
}
for name, f in runs.items():
    if ONLY and name not in ONLY:
        continue
    for _ in range(5):
        assert f() == 0
    torch.cuda.synchronize()
    evs = []
    for _ in range(40):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(1e3 * a.elapsed_time(b) for a, b in evs)
    nbytes = 4 * B * (3 * (Ho + 50) * (Wo + 50) + (4 if name.startswith("bwd") else 2) * 51 * Ho * Wo + 3 * Ho * Wo)
    print(json.dumps(dict(op=name, B=B, Ho=Ho, Wo=Wo, ns_per_pixel=round(1e3 * t[len(t) // 2] / (B * Ho * Wo), 4), mean_us=round(sum(t) / len(t), 1), min_us=round(t[0], 1), median_us=round(t[len(t) // 2], 1),
                          hbm_frac_median=None if name == "classify" else round(nbytes / t[len(t) // 2] / 1e6 / 8.0, 4),
                          errors=lib.savfi_sepconv_ws_errors())), flush=True)
