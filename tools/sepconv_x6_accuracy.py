"""Error of the sepconv filter-gradient kernels against a float64 evaluation (numpy), x6 vs the fp32-MFMA kernel (env)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meta_interpolation_amd import _hip
lib, st = _hip.lib(), _hip.current_stream()
B, C, Ho, Wo, K = 2, 3, 64, 64, 51
g = torch.Generator().manual_seed(3)
inp = torch.rand(B, C, Ho + K - 1, Wo + K - 1, generator=g)
v = torch.randn(B, K, Ho, Wo, generator=g) / 7
h = torch.randn(B, K, Ho, Wo, generator=g) / 7
gO = torch.randn(B, C, Ho, Wo, generator=g)
i64, v64, h64, g64 = (t.double().numpy() for t in (inp, v, h, gO))
gV = np.zeros((B, K, Ho, Wo)); gH = np.zeros((B, K, Ho, Wo))
for fy in range(K):
    for fx in range(K):
        win = i64[:, :, fy:fy + Ho, fx:fx + Wo]                    # [B,C,Ho,Wo]
        s = (win * g64).sum(1)                                     # sum_c gO * in
        gV[:, fy] += s * h64[:, fx]
        gH[:, fx] += s * v64[:, fy]
d = [t.cuda() for t in (inp, v, h, gO)]
oV, oH = torch.empty_like(d[1]), torch.empty_like(d[2])
_hip.check(lib.savfi_sepconv_bwd_f32(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), None, oV.data_ptr(), oH.data_ptr(), B, C, Ho, Wo, K, st), "bwd")
torch.cuda.synchronize()
for name, got, ref in (("gV", oV, gV), ("gH", oH, gH)):
    e = got.cpu().double().numpy() - ref
    print(name, "max|err| %.3e  rms err %.3e  mean err %.3e  max|ref| %.3f rms ref %.3f" % (np.abs(e).max(), np.sqrt((e ** 2).mean()), e.mean(), np.abs(ref).max(), np.sqrt((ref ** 2).mean())),
          "worst at", np.unravel_index(np.abs(e).argmax(), e.shape))
