"""Where a wino_conv3x3 workgroup spends its time: per-workgroup timestamps (s_memrealtime, 100 MHz) at kernel start, after the
prologue, after the channel loop, after the output stage and after its stores have been acknowledged, plus the CU it ran on.

Needs a library built with the trace hooks (they are compiled out of the product build):

    python tools/wino_trace.py --build          # -> tools/scratch/libsavfi_hip_trace.so
    SAVFI_HIP_LIB=tools/scratch/libsavfi_hip_trace.so python tools/wino_trace.py

Prints, per layer shape: the phase durations (mean / percentiles), how much of a CU's time had two / one / no workgroup
resident, and how many workgroups were in their output stage at the same time.
"""
import os, sys, ctypes, collections, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--build" in sys.argv:
    csrc = os.path.join(ROOT, "meta-interpolation_amd", "csrc")
    out = os.path.join(ROOT, "tools", "scratch", "libsavfi_hip_trace.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    srcs = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hip"))
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc", "-fno-slp-vectorize", "-DWINO_TRACE=1",
           "-I", os.path.join(ROOT, "include"), "-I", csrc] + srcs + ["-o", out]
    subprocess.check_call(cmd)
    print(out)
    sys.exit(0)
import numpy as np, torch
from meta_interpolation_amd import hip_ops
lib = ctypes.CDLL(os.path.abspath(os.environ["SAVFI_HIP_LIB"]))
lib.savfi_debug_wino_trace.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
for ci,co,h,w in [(32,32,384,512),(64,64,192,256),(128,128,96,128),(51,51,258,450)]:
    x = torch.randn(8,ci,h,w,device='cuda'); wt = torch.randn(co,ci,3,3,device='cuda')/(3*ci**.5); b = torch.randn(co,device='cuda')
    for _ in range(3): y = hip_ops.conv3x3(x,wt,b,0,0.0,1)
    torch.cuda.synchronize()
    buf = np.zeros((1 << 16, 8), dtype=np.uint64)
    n = lib.savfi_debug_wino_trace(buf.ctypes.data, 1 << 16)
    t = buf[:n].astype(np.int64)
    t0 = t[:, 0].min()
    us = lambda v: v / 100.0
    print("== %d->%d @%dx%d: %d workgroups, kernel span %.1f us" % (ci, co, h, w, n, us(t[:, 4].max() - t0)))
    for name, a, b_ in (("prologue", 0, 1), ("loop", 1, 2), ("output stage", 2, 3), ("store drain (waitcnt 0)", 3, 4), ("whole", 0, 4)):
        d = us(t[:, b_] - t[:, a]); print("   %-26s mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f us" % (name, d.mean(), *np.percentile(d, [10, 50, 90]), d.max()))
    hw = t[:, 5]; cu = (t[:, 6] & 0xf) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 0xf)
    busy2 = busy1 = idle = 0.0
    gaps = []
    for c in np.unique(cu):
        r = t[cu == c]; ev = sorted([(s, 1) for s in r[:, 0]] + [(e, -1) for e in r[:, 4]])
        lvl = 0; last = ev[0][0]
        for tm, dlt in ev:
            if lvl >= 2: busy2 += tm - last
            elif lvl == 1: busy1 += tm - last
            last = tm; lvl += dlt
        idle += (t[:, 4].max() - ev[-1][0]) + (ev[0][0] - t0)
    tot = busy2 + busy1 + idle
    print("   CUs seen %d; per-CU time with 2 WGs resident %.1f %%, 1 WG %.1f %%, 0 WG (head/tail) %.1f %%" % (len(np.unique(cu)), 100 * busy2 / tot, 100 * busy1 / tot, 100 * idle / tot))
    # phase: how many WGs are in the output stage / store drain at once
    ev = sorted([(s, 1) for s in t[:, 2]] + [(e, -1) for e in t[:, 4]]); lvl = 0; last = ev[0][0]; hist = collections.Counter()
    for tm, dlt in ev:
        hist[min(lvl // 64, 8)] += tm - last; last = tm; lvl += dlt
    tt = sum(hist.values()); print("   WGs simultaneously in output stage+drain (x64): " + " ".join("%d:%.0f%%" % (k, 100 * v / tt) for k, v in sorted(hist.items())))
