"""The wave-specialised SepConv filter-gradient kernel (csrc/sepconv_ws.hip) against the one-program-per-wave kernel
(csrc/sepconv_x6.hip: a variant library built with tools/build_variant.sh nows sepconv.hip -DSAVFI_SEPCONV_NO_WS, a second process) on the same seeded inputs, its protocol time-out counter, and
HIP-event timings of both at the bench shape.

    python tools/ws_check.py [--shapes 2x256x448,1x64x96,3x100x128] [--time-batch 8]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from meta_interpolation_amd import _hip  # noqa: E402

K, C = 51, 3


def inputs(B, Ho, Wo, seed=0):
    g = torch.Generator().manual_seed(seed)
    inp = torch.rand(B, C, Ho + K - 1, Wo + K - 1, generator=g)
    v = torch.randn(B, K, Ho, Wo, generator=g) / 7
    h = torch.randn(B, K, Ho, Wo, generator=g) / 7
    gO = torch.randn(B, C, Ho, Wo, generator=g)
    return [t.cuda() for t in (inp, v, h, gO)]


def run_bwd(B, Ho, Wo, iters=0):
    lib, st = _hip.lib(), _hip.current_stream()
    inp, v, h, gO = inputs(B, Ho, Wo)
    gV, gH = torch.full_like(v, float('nan')), torch.full_like(h, float('nan'))
    f = lambda: _hip.check(lib.savfi_sepconv_bwd_f32(inp.data_ptr(), v.data_ptr(), h.data_ptr(), gO.data_ptr(), None,
                                                     gV.data_ptr(), gH.data_ptr(), B, C, Ho, Wo, K, st), "bwd")
    f()
    torch.cuda.synchronize()
    us = None
    if iters:
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        evs = []
        for _ in range(iters):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f(); b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        t = sorted(1e3 * a.elapsed_time(b) for a, b in evs)
        us = dict(mean=sum(t) / len(t), min=t[0], median=t[len(t) // 2])
    return gV, gH, us


def run_fwd(B, Ho, Wo, iters=0):
    lib, st = _hip.lib(), _hip.current_stream()
    inp, v, h, _ = inputs(B, Ho, Wo)
    out = torch.full((B, C, Ho, Wo), float('nan'), device='cuda')
    f = lambda: _hip.check(lib.savfi_sepconv_fwd_f32(inp.data_ptr(), v.data_ptr(), h.data_ptr(), out.data_ptr(), B, C, Ho, Wo, K, st), "fwd")
    f()
    torch.cuda.synchronize()
    us = None
    if iters:
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        evs = []
        for _ in range(iters):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f(); b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        t = sorted(1e3 * a.elapsed_time(b) for a, b in evs)
        us = dict(mean=sum(t) / len(t), min=t[0], median=t[len(t) // 2])
    return out, us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shapes', default='1x64x96,2x256x448,3x100x128,1x8x32')
    ap.add_argument('--time-batch', type=int, default=8)
    ap.add_argument('--dump', default=None)
    ap.add_argument('--iters', type=int, default=30)
    o = ap.parse_args()
    shapes = [tuple(int(x) for x in s.split('x')) for s in o.shapes.split(',')]
    if o.dump:            # child: the other kernel's outputs
        for B, Ho, Wo in shapes:
            gV, gH, _ = run_bwd(B, Ho, Wo)
            np.save(os.path.join(o.dump, 'gV_%dx%dx%d.npy' % (B, Ho, Wo)), gV.cpu().numpy())
            np.save(os.path.join(o.dump, 'gH_%dx%dx%d.npy' % (B, Ho, Wo)), gH.cpu().numpy())
            np.save(os.path.join(o.dump, 'out_%dx%dx%d.npy' % (B, Ho, Wo)), run_fwd(B, Ho, Wo)[0].cpu().numpy())
        if o.time_batch:
            _, _, us = run_bwd(o.time_batch, 256, 448, iters=o.iters)
            print(json.dumps(dict(kernel='x6 bwd', B=o.time_batch, us=us)), flush=True)
            print(json.dumps(dict(kernel='x6 fwd', B=o.time_batch, us=run_fwd(o.time_batch, 256, 448, iters=o.iters)[1])), flush=True)
        return
    tmp = tempfile.mkdtemp(prefix='ws_check_')
    env = dict(os.environ, SAVFI_HIP_LIB=os.path.join(os.path.dirname(os.path.abspath(__file__)), 'variants', 'libsavfi_nows.so'))
    child = subprocess.run([sys.executable, os.path.abspath(__file__), '--shapes', o.shapes, '--dump', tmp,
                            '--time-batch', str(o.time_batch), '--iters', str(o.iters)], env=env, capture_output=True, text=True)
    print(child.stdout.strip(), flush=True)
    if child.returncode:
        print(child.stderr[-2000:])
    lib = _hip.lib()
    ok = True
    for B, Ho, Wo in shapes:
        gV, gH, _ = run_bwd(B, Ho, Wo)
        errs = lib.savfi_sepconv_ws_errors()
        rV, rH = np.load(os.path.join(tmp, 'gV_%dx%dx%d.npy' % (B, Ho, Wo))), np.load(os.path.join(tmp, 'gH_%dx%dx%d.npy' % (B, Ho, Wo)))
        dV = np.abs(gV.cpu().numpy() - rV)
        dH = np.abs(gH.cpu().numpy() - rH)
        rec = dict(shape=[B, Ho, Wo], ws_errors=errs, gV_max=float(np.nanmax(dV)), gH_max=float(np.nanmax(dH)),
                   gV_nan=int(np.isnan(dV).sum()), gH_nan=int(np.isnan(dH).sum()), gV_scale=float(np.abs(rV).max()), gH_scale=float(np.abs(rH).max()))
        rO = np.load(os.path.join(tmp, 'out_%dx%dx%d.npy' % (B, Ho, Wo)))
        dO = np.abs(run_fwd(B, Ho, Wo)[0].cpu().numpy() - rO)
        rec.update(out_max=float(np.nanmax(dO)), out_nan=int(np.isnan(dO).sum()), out_scale=float(np.abs(rO).max()), ws_errors=lib.savfi_sepconv_ws_errors())
        if rec['out_nan'] or rec['out_max'] > 1e-5 * rec['out_scale']:
            ok = False
            io = np.argwhere(~(dO <= 1e-5 * rec['out_scale']))
            rec['out_bad'] = [len(io)] + io[:6].tolist()
        bad = rec['gV_nan'] or rec['gH_nan'] or rec['gV_max'] > 1e-5 * rec['gV_scale'] or rec['gH_max'] > 1e-5 * rec['gH_scale'] or errs
        if bad:
            ok = False
            iv = np.argwhere(~(dV <= 1e-5 * rec['gV_scale']))
            ih = np.argwhere(~(dH <= 1e-5 * rec['gH_scale']))
            rec['gV_bad'] = [len(iv)] + iv[:6].tolist()
            rec['gH_bad'] = [len(ih)] + ih[:6].tolist()
        print(json.dumps(rec), flush=True)
    if o.time_batch:
        _, _, us = run_bwd(o.time_batch, 256, 448, iters=o.iters)
        nbytes = 4 * o.time_batch * (3 * 306 * 498 + 4 * 51 * 256 * 448 + 3 * 256 * 448)
        print(json.dumps(dict(kernel='ws fwd', B=o.time_batch, us=run_fwd(o.time_batch, 256, 448, iters=o.iters)[1])), flush=True)
        print(json.dumps(dict(kernel='ws bwd', B=o.time_batch, us=us, hbm_frac_mean=nbytes / us['mean'] / 1e6 / 8000.0, ws_errors=lib.savfi_sepconv_ws_errors())), flush=True)
    print('OK' if ok else 'MISMATCH', flush=True)


if __name__ == '__main__':
    main()
