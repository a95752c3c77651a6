"""Driver for the HBM-traffic PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs).

    python tools/hbm_traffic.py run            # the workload that is profiled (calibration copy + sepconv fwd/bwd)
    python tools/hbm_traffic.py parse DIR_FETCH DIR_WRITE > profiles/rXX_hbm_traffic.json

Calibration (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE reads exactly half the bytes of a wide
coalesced stream and WRITE_SIZE is uncalibrated, so a 512 MiB float4 device copy of known size is profiled in
the same pass and gives the per-counter correction factor that is applied to the sepconv kernels.
"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CAL_BYTES = 512 * 1024 * 1024
# (B, Ho, Wo): the full padded canvas of the reference and the frame window this build evaluates (sepconv/model.py)
CASES = [(1, 384, 512), (1, 256, 448), (2, 256, 448), (4, 256, 448), (8, 256, 448)]     # B = 8 / 4: lockstep support / target pass
# (B, Ho, Wo, pair): the pair launches of the plugin's tail (savfi_sepconv_{fwd,bwd}_pair_frames8_f32: both local convolutions of B samples)
PAIR_CASES = [(4, 256, 448), (8, 256, 448)]


def mfma_rows(B, Ho, Wo):
    """csrc/sepconv.hip mfma_rows(): rows per workgroup of the MFMA kernels."""
    best, best_cost = 12, None
    for r in (12, 8, 16):
        wgs = B * -(-Wo // 64) * -(-Ho // r)
        cost = -(-wgs // 256) * (r + 2)
        if best_cost is None or cost < best_cost:
            best, best_cost = r, cost
    return best


def run():
    import torch
    from meta_interpolation_amd import _hip
    from meta_interpolation_amd.sepconv.sepconv_op.sepconv import algorithmic_bytes  # noqa: F401
    lib, st = _hip.lib(), _hip.current_stream()
    a = torch.empty(CAL_BYTES // 4, device='cuda').normal_()
    bb = torch.empty_like(a)
    for _ in range(3):
        bb.copy_(a)
    for (B, Ho, Wo) in CASES:
        C, K = 3, 51
        # frames of 8-bit images (k / 255), through the entry points the plugin uses: per call the three-product kernel and the
        # six-product instance's early exit (both launched, the device picks: csrc/sepconv_ws.hip)
        inp = torch.randint(0, 256, (B, C, Ho + K - 1, Wo + K - 1), device='cuda').float().div(255)
        words = torch.empty(256, dtype=torch.int32, device='cuda')
        lib.savfi_frames8_classify_f32(inp.data_ptr(), inp.numel(), words.data_ptr(), st)
        v = torch.randn(B, K, Ho, Wo, device='cuda') / 7
        h = torch.randn(B, K, Ho, Wo, device='cuda') / 7
        gO = torch.randn(B, C, Ho, Wo, device='cuda')
        out, gV, gH = torch.empty_like(gO), torch.empty_like(v), torch.empty_like(h)
        for _ in range(3):
            assert lib.savfi_sepconv_fwd_frames8_f32(inp.data_ptr(), v.data_ptr(), h.data_ptr(), out.data_ptr(), words.data_ptr(), B, C, Ho, Wo, K, K, 1, st) == 0      # taps unit-major, as the plugin calls it
            assert lib.savfi_sepconv_bwd_frames8_f32(inp.data_ptr(), v.data_ptr(), h.data_ptr(), gO.data_ptr(), gV.data_ptr(), gH.data_ptr(),
                                                     words.data_ptr(), B, C, Ho, Wo, K, K, 3, st) == 0   # taps and gradients unit-major (inner loop)
        torch.cuda.synchronize()
    for (B, Ho, Wo) in PAIR_CASES:
        C, K = 3, 51
        fr = [torch.randint(0, 256, (B, C, Ho + K - 1, Wo + K - 1), device='cuda').float().div(255) for _ in range(2)]
        words = [torch.empty(256, dtype=torch.int32, device='cuda') for _ in range(2)]
        for f, wd in zip(fr, words):
            lib.savfi_frames8_classify_f32(f.data_ptr(), f.numel(), wd.data_ptr(), st)
        taps = torch.randn(4 * B, K, Ho, Wo, device='cuda') / 7
        gO = torch.randn(B, C, Ho, Wo, device='cuda')
        out2, gT = torch.empty(B, 2, C, Ho, Wo, device='cuda'), torch.empty_like(taps)
        for _ in range(3):
            assert lib.savfi_sepconv_fwd_pair_frames8_f32(fr[0].data_ptr(), fr[1].data_ptr(), taps.data_ptr(), out2.data_ptr(), words[0].data_ptr(),
                                                          words[1].data_ptr(), B, C, Ho, Wo, K, 1, st) == 0
            assert lib.savfi_sepconv_bwd_pair_frames8_f32(fr[0].data_ptr(), fr[1].data_ptr(), taps.data_ptr(), gO.data_ptr(), gT.data_ptr(),
                                                          words[0].data_ptr(), words[1].data_ptr(), B, C, Ho, Wo, K, 3, st) == 0
        torch.cuda.synchronize()


def _collect(d, counter):
    """Counter values per sepconv launch, keyed (kernel kind, case index): run() launches, case after case, three fwd/bwd pairs,
    so the n-th sepconv_fwd (sepconv_bwd) dispatch belongs to case n // 3 -- the persistent backward kernel always launches one
    workgroup per CU, its grid no longer identifies the shape."""
    path = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith('counter_collection.csv')][0]
    rows = [r for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    out, seen = {}, {'sepconv_fwd': 0, 'sepconv_bwd': 0}
    first_sepconv = min((int(r['Dispatch_Id']) for r in rows if 'sepconv' in r['Kernel_Name']), default=1 << 62)
    for row in rows:
        name = row['Kernel_Name']
        if 'elementwise' in name or 'copy' in name.lower():
            # the calibration copies run before the first sepconv launch; later cases allocate tensors larger than the calibration buffer
            if int(row['Dispatch_Id']) < first_sepconv:
                out.setdefault(('copy', -1), []).append(float(row['Counter_Value']))
            continue
        kind = 'sepconv_fwd' if 'sepconv_fwd' in name else 'sepconv_bwd' if 'sepconv_bwd' in name else None
        if kind is None:
            continue
        out.setdefault((kind, seen[kind] // 6), []).append(float(row['Counter_Value']))       # two dispatches per call: see run()
        seen[kind] += 1
    # the calibration copy is the largest of the elementwise / copy kernels (tensor initialisation launches some small ones)
    return {k: (max(v) if k[0] == 'copy' else sum(v) / 3.0) for k, v in out.items()}


def parse(dfetch, dwrite):
    from meta_interpolation_amd.sepconv.sepconv_op.sepconv import algorithmic_bytes
    f, w = _collect(dfetch, 'FETCH_SIZE'), _collect(dwrite, 'WRITE_SIZE')
    cal_f, cal_w = f[('copy', -1)], w[('copy', -1)]
    # counters are in KiB; correction = true bytes / reported bytes on the known copy
    corr_f = CAL_BYTES / (cal_f * 1024.0)
    corr_w = CAL_BYTES / (cal_w * 1024.0)
    res = {"unit": "bytes per launch", "calibration": {"copy_bytes_each_way": CAL_BYTES, "FETCH_SIZE_KiB": cal_f,
                                                       "WRITE_SIZE_KiB": cal_w, "fetch_correction": corr_f,
                                                       "write_correction": corr_w}, "kernels": {}}
    for (kind, ci), val in sorted(f.items()):
        if kind == 'copy' or ci >= len(CASES) + len(PAIR_CASES):
            continue
        wv = w.get((kind, ci), 0.0)
        pair = ci >= len(CASES)
        B, Ho, Wo = PAIR_CASES[ci - len(CASES)] if pair else CASES[ci]
        alg = (2 if pair else 1) * algorithmic_bytes(B, 3, Ho, Wo, 51, grads=2 if kind == 'sepconv_bwd' else 0)
        rd, wr = val * 1024 * corr_f, wv * 1024 * corr_w
        res["kernels"]["%s_%sB%d_%dx%d" % (kind, "pair" if pair else "", B, Ho, Wo)] = {
            "FETCH_SIZE_KiB": val, "WRITE_SIZE_KiB": wv, "hbm_read_bytes": rd, "hbm_write_bytes": wr, "traffic": rd + wr,
            "algorithmic_bytes": alg, "traffic_over_algorithmic": (rd + wr) / alg}
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run()
    else:
        parse(sys.argv[2], sys.argv[3])
