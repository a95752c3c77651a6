"""hipMemsetAsync captured into a hipGraph (torch.cuda.CUDAGraph) and replayed three times: the buffer is dirtied before every replay,
a kernel inside the graph copies it right after the memset.  On ROCm 7.2 / torch 2.10+rocm7.0 only the FIRST replay clears (see
profiles/r03_graph_memset_nodes.txt): ATen reductions that pick several workgroups per output clear their semaphores this way.
"""
import ctypes, sys, torch
hip = ctypes.CDLL('libamdhip64.so')
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
def run(nbytes, lead, trail):
    n = nbytes // 4
    buf = torch.full((max(n, 1),), 5, dtype=torch.int32, device='cuda'); out = torch.empty_like(buf); z = torch.zeros(8, device='cuda')
    src = torch.arange(max(n, 1), dtype=torch.int32, device='cuda') + 100
    def body():
        if lead: z.add_(1.0)
        st = torch.cuda.current_stream().cuda_stream
        rc = hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, nbytes, ctypes.c_void_p(st)); assert rc == 0, rc
        out.copy_(buf)                # kernel that reads what the memset wrote
        if trail: buf.copy_(src)      # a later kernel dirties the buffer again (as a reused pool block would)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    res = []
    for it in range(3):
        buf.fill_(7 + it); out.fill_(-1); torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        res.append((int((out != 0).sum()), out[:3].tolist()))
    print("memset %7d B lead=%d trail=%d: nonzero after replay" % (nbytes, lead, trail), res, flush=True)
for nb in (4, 60, 1024, 15360, 1 << 20):
    for lead in (0, 1):
        for trail in (0, 1):
            run(nb, lead, trail)

# ---- the ATen reduction that depends on such a node: x.mean(2) of a 3 x 720 x 1280 frame (several workgroups per output, the
# last one -- found through the memset semaphores -- writes the result), followed by .mean(3) as in the reference's sub_mean ---------
x = torch.rand(1, 3, 720, 1280, device='cuda')
def two_stage():
    a = x.mean(2, keepdim=True)
    return [a, a.mean(3, keepdim=True)]
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2): two_stage()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    outs = two_stage()
for it in range(4):
    if it: x.uniform_(0.0, 1.0 + it)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    got = [o.clone() for o in outs]
    want = two_stage(); torch.cuda.synchronize()
    print("captured x.mean(2).mean(3), replay %d (%s frame): max |graph - eager| = %s" % (
        it, "new" if it else "captured", ["%.2e" % float((a - b).abs().max()) for a, b in zip(got, want)]), flush=True)
