"""Section trace of sepconv_bwd_ws (library built with -DWS_TRACE=1, SAVFI_HIP_LIB): wave cycles per section of workgroup 0."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meta_interpolation_amd import _hip
B, C, Ho, Wo, K = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 3, 256, 448, 51
lib, st = _hip.lib(), _hip.current_stream()
U16 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
F8 = len(sys.argv) > 2 and sys.argv[2] == "f8"       # frames of 8-bit images through savfi_sepconv_bwd_frames8_f32 (the three-product kernel)
inp = torch.randint(0, 256, (B, C, Ho + K - 1, Wo + K - 1), device="cuda").float().div(255) if F8 else torch.rand(B, C, Ho + K - 1, Wo + K - 1, device="cuda")
v = torch.randn(B, K, Ho, Wo, device="cuda") / 7
h = torch.randn(B, K, Ho, Wo, device="cuda") / 7
gO = torch.randn(B, C, Ho, Wo, device="cuda")
gV, gH = torch.empty_like(v), torch.empty_like(h)
if F8:
    from meta_interpolation_amd.sepconv.sepconv_op import sepconv as S
    words = S.frames8_classify(inp)
    f = lambda: _hip.check(lib.savfi_sepconv_bwd_frames8_f32(inp.data_ptr(), v.data_ptr(), h.data_ptr(), gO.data_ptr(), gV.data_ptr(), gH.data_ptr(), words.data_ptr(), B, C, Ho, Wo, K, K, U16, st), "bwd8")
else:
    f = lambda: _hip.check(lib.savfi_sepconv_bwd_f32(inp.data_ptr(), v.data_ptr(), h.data_ptr(), gO.data_ptr(), None, gV.data_ptr(), gH.data_ptr(), B, C, Ho, Wo, K, st), "bwd")
buf = (ctypes.c_ulonglong * 256)()
lib.savfi_sepconv_ws_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
for _ in range(3): f()
lib.savfi_sepconv_ws_trace(buf, 1)
wgbuf = (ctypes.c_ulonglong * 2048)()
lib.savfi_sepconv_ws_trace_wg.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.savfi_sepconv_ws_trace_wg(wgbuf, 1)
NL = 5
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(NL): f()
e1.record(); torch.cuda.synchronize()
print("traced build: %.1f us per launch" % (1e3 * e0.elapsed_time(e1) / NL))
lib.savfi_sepconv_ws_trace(buf, 1)
units = 2 * ((B * 14 * 64 + 255) // 256)
print("per unit (2 passes) cycles, workgroup 0, %d units per pair, lib %s" % (units, os.environ.get("SAVFI_HIP_LIB")))
MF = ["top", "-", "-", "gV MFMA loop", "wait tab(v) + v fragments + first gH A fragments", "gV scale + wait out_free", "tile write+set", "-", "-", "gH MFMA loop", "next rows + first gV A fragments", "gH scale + wait out_free", "tile write+set", "wait tab(h next)", "h fragments + slide wait"]
SG_OLD = ["top: granule loads, readlanes", "wait tab_free", "h table write", "B: side, tails-a, h loads", "wait out_full(gH)", "drain gH", "gV tail sums", "wait tab_free", "v table write", "E: gH tails, v loads", "wait out_full(gV)", "drain gV", "wait prog", "granule write"]
SG = ["top: (slot reads), row load, readlanes", "wait tab_free", "table write+set", "tails (side, sums), tap loads", "wait prog", "window row write", "wait out_full(prev)", "drain+stores(prev)", "v: slide wait + side reads", "v: tail sums", "DMA: fetch issue (7 instr)", "DMA: vmcnt wait"]
two = os.environ.get('SAVFI_SEPCONV_WS2') is not None
MF2 = ['top', '-', '-', 'gV half 1 loop (72 MFMAs)', 'tail wait + epilogue 1 (2 stores)', 'gV half 2 loop (48)', 'v frags + epilogue 2 (2 stores)', '-', '-', 'gH half 1 loop (72)', 'gH half 2 loop (72)', 'epilogue h1 (wait out_free, 8 writes)', 'next h frags, rows, epilogue h2, set', '-', '-']
for w in range(16 if two else 12):
    row = [buf[w * 16 + k] / NL / units for k in range(16)]
    names = (MF2 if w < 8 else SG) if two else (MF if w < 4 else SG)
    print("wave %d kernel cycles per launch %.0f; " % (w, buf[w * 16 + 15] / NL), end="")
    row[15] = 0
    print("wave %d total %.0f: " % (w, sum(row)) + " | ".join("%s %.0f" % (names[k], row[k]) for k in range(len(names))))

# per workgroup: cycles from entry to exit (wave 0), by number of runs
lib.savfi_sepconv_ws_trace_wg(wgbuf, 1)
rows = [(wgbuf[2 * i] / NL, wgbuf[2 * i + 1] / NL) for i in range(1024) if wgbuf[2 * i]]
if rows:
    for nr in sorted(set(round(r[1]) for r in rows)):
        c = sorted(r[0] for r in rows if round(r[1]) == nr)
        print("workgroups with %d run(s): %d, cycles min %.0f median %.0f max %.0f" % (nr, len(c), c[0], c[len(c) // 2], c[-1]))
    c = sorted(r[0] for r in rows)
    print("all %d workgroups: min %.0f mean %.0f max %.0f (max / mean %.3f)" % (len(c), c[0], sum(c) / len(c), c[-1], c[-1] / (sum(c) / len(c))))
    for x in range(8):
        c = [r[0] for i, r in enumerate(rows) if i % 8 == x]
        print("XCD %d (blockIdx %% 8): %d workgroups, mean %.0f min %.0f max %.0f" % (x, len(c), sum(c) / len(c), min(c), max(c)))
