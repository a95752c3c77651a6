import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from meta_interpolation_amd import _hip
from meta_interpolation_amd.sepconv.sepconv_op import sepconv as S
from oracle import torch_ops as O
K = 51
def run(B, Ho, Wo, seed):
    g = torch.Generator().manual_seed(seed)
    inp = torch.randint(0, 256, (B, 3, Ho + K - 1, Wo + K - 1), generator=g).float().div(255)
    v = torch.randn(B, K, Ho, Wo, generator=g) / 7
    h = torch.randn(B, K, Ho, Wo, generator=g) / 7
    gO = torch.randn(B, 3, Ho, Wo, generator=g)
    _, rV, rH = O.sepconv_backward_c(inp, v, h, gO)
    lib, st = _hip.lib(), _hip.current_stream()
    di, dv, dh, dg = inp.cuda(), v.cuda(), h.cuda(), gO.cuda()
    words = S.frames8_classify(di)
    for rep in range(3):
        gV, gH = torch.full_like(dv, float('nan')), torch.full_like(dh, float('nan'))
        rc = lib.savfi_sepconv_bwd_frames8_f32(di.data_ptr(), dv.data_ptr(), dh.data_ptr(), dg.data_ptr(), gV.data_ptr(), gH.data_ptr(), words.data_ptr(), B, 3, Ho, Wo, K, K, 0, st)
        torch.cuda.synchronize()
        for name, got, ref in (("gV", gV.cpu(), rV), ("gH", gH.cpu(), rH)):
            d = (got - ref).abs()
            d[torch.isnan(d)] = 1e9
            bad = (d > 1e-4 * ref.abs().max()).nonzero()
            print(B, Ho, Wo, "rep", rep, name, "rc", rc, "bad", len(bad), "first", bad[:6].tolist(), "err", lib.savfi_sepconv_ws_errors())
for shp in ((1, 16, 32), (2, 37, 36), (2, 64, 96)):
    run(*shp, seed=100 * shp[0] + shp[1])
