"""HIP-event timing of savfi_sepconv_bwd_f32 at B x 256 x 448 for the library SAVFI_HIP_LIB names (kernel variants)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meta_interpolation_amd import _hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
C, Ho, Wo, K = 3, 256, 448, 51
lib, st = _hip.lib(), _hip.current_stream()
inp = torch.rand(B, C, Ho + K - 1, Wo + K - 1, device="cuda")
v = torch.randn(B, K, Ho, Wo, device="cuda") / 7
h = torch.randn(B, K, Ho, Wo, device="cuda") / 7
gO = torch.randn(B, C, Ho, Wo, device="cuda")
gV, gH = torch.empty_like(v), torch.empty_like(h)
f = lambda: _hip.check(lib.savfi_sepconv_bwd_f32(inp.data_ptr(), v.data_ptr(), h.data_ptr(), gO.data_ptr(), None, gV.data_ptr(), gH.data_ptr(), B, C, Ho, Wo, K, st), "bwd")
for _ in range(5): f()
torch.cuda.synchronize()
evs = []
for _ in range(40):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f(); b.record(); evs.append((a, b))
torch.cuda.synchronize()
t = sorted(1e3 * a.elapsed_time(b) for a, b in evs)
nbytes = 4 * B * (3 * 306 * 498 + 4 * 51 * 256 * 448 + 3 * 256 * 448)
print(json.dumps(dict(lib=os.path.basename(os.environ.get("SAVFI_HIP_LIB", "default")), B=B, mean_us=round(sum(t) / len(t), 1), min_us=round(t[0], 1), median_us=round(t[len(t) // 2], 1),
                      hbm_frac_median=round(nbytes / t[len(t) // 2] / 1e6 / 8.0, 4), errors=lib.savfi_sepconv_ws_errors())), flush=True)
