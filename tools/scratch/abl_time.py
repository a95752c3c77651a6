import ctypes, glob, os, sys, torch
B, C, Ho, Wo, K = 1, 3, 384, 512, 51
inp = torch.rand(B, C, Ho + K - 1, Wo + K - 1, device='cuda'); v = torch.randn(B, K, Ho, Wo, device='cuda') / 7
h = torch.randn(B, K, Ho, Wo, device='cuda') / 7; out = torch.empty(B, C, Ho, Wo, device='cuda')
gO = torch.randn(B, C, Ho, Wo, device='cuda'); gV, gH = torch.empty_like(v), torch.empty_like(h)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = ctypes.c_void_p
for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'abl', '*.so'))):
    lib = ctypes.CDLL(path)
    f = lib.savfi_sepconv_fwd_f32; f.argtypes = [P, P, P, P] + [ctypes.c_int] * 5 + [P]
    g = lib.savfi_sepconv_bwd_f32; g.argtypes = [P] * 7 + [ctypes.c_int] * 5 + [P]
    def run_f(): f(inp.data_ptr(), v.data_ptr(), h.data_ptr(), out.data_ptr(), B, C, Ho, Wo, K, st)
    def run_b(): g(inp.data_ptr(), v.data_ptr(), h.data_ptr(), gO.data_ptr(), None, gV.data_ptr(), gH.data_ptr(), B, C, Ho, Wo, K, st)
    res = []
    for fn in (run_f, run_b):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): fn()
        b.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / 20 * 1e3)
    print('%-34s fwd %.1f us   bwd %.1f us' % (os.path.basename(path), res[0], res[1]))
