"""Section trace of sepconv_fwd_ws<U8> (library built with -DWS_TRACE=1, SAVFI_HIP_LIB): wave cycles per section of workgroup 0, and the
per-workgroup cycle distribution.  python tools/ws_trace_fwd.py [B] [unit16]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meta_interpolation_amd import _hip
from meta_interpolation_amd.sepconv.sepconv_op import sepconv as S
B, C, Ho, Wo, K = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 3, 256, 448, 51
U16 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib, st = _hip.lib(), _hip.current_stream()
inp = torch.randint(0, 256, (B, C, Ho + K - 1, Wo + K - 1), device="cuda").float().div(255)
v = torch.randn(B, K, Ho, Wo, device="cuda") / 7
h = torch.randn(B, K, Ho, Wo, device="cuda") / 7
out = torch.empty(B, C, Ho, Wo, device="cuda")
words = S.frames8_classify(inp)
f = lambda: _hip.check(lib.savfi_sepconv_fwd_frames8_f32(inp.data_ptr(), v.data_ptr(), h.data_ptr(), out.data_ptr(), words.data_ptr(), B, C, Ho, Wo, K, K, U16, st), "fwd8")
buf, wgbuf = (ctypes.c_ulonglong * 256)(), (ctypes.c_ulonglong * 2048)()
lib.savfi_sepconv_ws_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.savfi_sepconv_ws_trace_wg.argtypes = [ctypes.c_void_p, ctypes.c_int]
for _ in range(3): f()
lib.savfi_sepconv_ws_trace(buf, 1); lib.savfi_sepconv_ws_trace_wg(wgbuf, 1)
NL = 5
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(NL): f()
e1.record(); torch.cuda.synchronize()
print("traced build: %.1f us per launch" % (1e3 * e0.elapsed_time(e1) / NL))
lib.savfi_sepconv_ws_trace(buf, 1)
units = 2 * ((B * 14 * 64 + 255) // 256)
MF = ["top", "MFMA loop (60)", "wait tab(h next) + h frags + rows + first A frags", "wait v tile", "vertical pass", "wait op_free", "op write+set"]
SH = ["top: row loads", "wait tab_free", "table write+set", "tap loads (issue)", "wait prog", "window row write", "wait op/tl full(prev)", "final add + store(prev)", "-", "-", "-", "-"]
SV = ["top: row loads, hraw", "wait vt_free", "v tile write+set", "tap loads (issue)", "wait prog", "window row write", "-", "-", "slide wait + side reads", "tail sums (6 wave sums)", "wait tl_free", "tl write"]
for w in range(12):
    row = [buf[w * 16 + k] / NL / units for k in range(16)]
    names = MF if w < 4 else SH if w < 8 else SV
    tot = buf[w * 16 + 15] / NL
    row[15] = 0
    print("wave %d kernel cycles %.0f; per unit %.0f: " % (w, tot, sum(row)) + " | ".join("%s %.0f" % (names[k], row[k]) for k in range(len(names)) if names[k] != "-"))
lib.savfi_sepconv_ws_trace_wg(wgbuf, 1)
rows = [(wgbuf[2 * i] / NL, wgbuf[2 * i + 1] / NL) for i in range(1024) if wgbuf[2 * i]]
c = sorted(r[0] for r in rows)
print("all %d workgroups: min %.0f mean %.0f max %.0f (max / mean %.3f)" % (len(c), c[0], sum(c) / len(c), c[-1], c[-1] / (sum(c) / len(c))))
