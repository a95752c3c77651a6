"""raw ctypes A/B of the U8 backward on planar taps: python tools/r5/dbg2.py LIB"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import torch_ops as O
K = 51
lib = ctypes.CDLL(sys.argv[1])
P = ctypes.c_void_p
lib.savfi_sepconv_bwd_frames8_f32.argtypes = [P] * 7 + [ctypes.c_int] * 7 + [P]
lib.savfi_frames8_classify_f32.argtypes = [P, ctypes.c_int64, P, P]
def run(B, Ho, Wo, seed):
    g = torch.Generator().manual_seed(seed)
    inp = torch.randint(0, 256, (B, 3, Ho + K - 1, Wo + K - 1), generator=g).float().div(255)
    v = torch.randn(B, K, Ho, Wo, generator=g) / 7
    h = torch.randn(B, K, Ho, Wo, generator=g) / 7
    gO = torch.randn(B, 3, Ho, Wo, generator=g)
    _, rV, rH = O.sepconv_backward_c(inp, v, h, gO)
    di, dv, dh, dg = inp.cuda(), v.cuda(), h.cuda(), gO.cuda()
    words = torch.empty(256, dtype=torch.int32, device='cuda')
    assert lib.savfi_frames8_classify_f32(di.data_ptr(), di.numel(), words.data_ptr(), None) == 0
    nbad = 0
    for rep in range(10):
        gV, gH = torch.full_like(dv, float('nan')), torch.full_like(dh, float('nan'))
        torch.cuda.synchronize()
        rc = lib.savfi_sepconv_bwd_frames8_f32(di.data_ptr(), dv.data_ptr(), dh.data_ptr(), dg.data_ptr(), gV.data_ptr(), gH.data_ptr(), words.data_ptr(), B, 3, Ho, Wo, K, K, 0, None)
        torch.cuda.synchronize()
        for name, got, ref in (("gV", gV.cpu(), rV), ("gH", gH.cpu(), rH)):
            d = (got - ref).abs()
            d[torch.isnan(d)] = 1e9
            bad = (d > 1e-4 * ref.abs().max()).nonzero()
            nbad += len(bad)
            if len(bad) and rep < 2:
                print(B, Ho, Wo, "rep", rep, name, "rc", rc, "bad", len(bad), "taps", sorted(set(bad[:, 1].tolist())), "x%4", sorted(set((bad[:, 3] % 4).tolist())))
    print(os.path.basename(sys.argv[1]), B, Ho, Wo, "bad elements over 10 reps:", nbad)
for shp in ((1, 16, 32), (2, 37, 36), (2, 64, 96)):
    run(*shp, seed=100 * shp[0] + shp[1])
