import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from meta_interpolation_amd import _hip
lib = _hip.lib()
dev = torch.device('cuda')
random.seed(2); torch.manual_seed(2)
G = 1 << 16   # guard floats on each side
bad = 0
shapes = [(2, 32, 32, 384, 512, 1, 0), (2, 6, 32, 384, 512, 1, 0), (2, 51, 51, 258, 450, 0, 0), (2, 51, 51, 256, 448, 0, 1), (1, 64, 64, 192, 256, 1, 0),
          (2, 64, 51, 136, 233, 1, 0), (2, 64, 64, 136, 233, 1, 1), (1, 32, 32, 384, 512, 1, 1), (2, 32, 6, 384, 512, 1, 1)]
for it in range(120):
    if it < len(shapes):
        n, ci, co, h, w, pad, mode = shapes[it]
    else:
        n = random.choice([1, 2, 3]); ci = random.choice([3, 6, 8, 32, 51, 64, 128]); co = random.choice([3, 8, 32, 51, 64, 128])
        h = random.choice([5, 31, 64, 130, 258]); w = random.choice([4, 7, 33, 100, 233, 450]); pad = random.choice([0, 1]); mode = random.choice([0, 1])
        if pad == 0 and (h < 3 or w < 3):
            continue
    K, I = (ci, co) if mode == 0 else (co, ci)
    grow = 2 * (pad if mode == 0 else 2 - pad) - 2
    ho, wo = h + grow, w + grow
    x = torch.randn(n, K, h, w, device=dev)
    wt = torch.randn(co, ci, 3, 3, device=dev) / 10
    b = torch.randn(co, device=dev)
    nws = int(lib.savfi_conv3x3_workspace_floats(n, ci, co, h, w, pad, mode))
    nout = n * I * ho * wo
    wsb = torch.full((nws + 2 * G,), 7.25, device=dev)
    outb = torch.full((nout + 2 * G,), 7.25, device=dev)
    rc = lib.savfi_conv3x3_f32(x.data_ptr(), wt.data_ptr(), b.data_ptr() if mode == 0 else None, outb.data_ptr() + 4 * G, wsb.data_ptr() + 4 * G,
                               n, ci, co, h, w, pad, mode, 1.0, _hip.current_stream())
    assert rc == 0, rc
    torch.cuda.synchronize()
    g_ok = bool((wsb[:G] == 7.25).all() and (wsb[G + nws:] == 7.25).all() and (outb[:G] == 7.25).all() and (outb[G + nout:] == 7.25).all())
    full = bool((outb[G:G + nout] != 7.25).all())
    if not g_ok or not full:
        bad += 1
        print("BAD", (n, ci, co, h, w, pad, mode), "guards ok", g_ok, "all written", full, flush=True)
print("done bad =", bad)
