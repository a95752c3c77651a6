"""oracle/gen_sensitivity.py -- TEST INFRASTRUCTURE.  Measures the REFERENCE's own numerical spread.

    python oracle/gen_sensitivity.py [--only case ...]      # writes tests/golden/sensitivity.npz

Why.  North star: outputs within 1e-4 pixel L1 / 1e-3 dB of the reference's CPU path.  For SGD-type inner
rules the whole path is a smooth function of the convolution outputs and the HIP path meets those bounds
with two orders of magnitude to spare.  Adam / Adamax-type rules are not smooth: an update is
+-lr*c per element whatever |g| (g/(|g|+1e-8), m/(sqrt(v)+1e-8)), so an element whose gradient is below the
rounding noise of a convolution steps the other way under ANY other summation order -- including the
reference's own CUDA kernels vs its CPU kernels.  This script measures that: it runs the imported reference
(same shims as gen_golden.py, no reference file touched) on the fixture cases

    base : float32, as in gen_golden.py                    (must reproduce tests/golden/system_<case>.npz)
    perm : float32, every F.conv2d evaluated on channel-reversed, horizontally flipped operands
           (mathematically the same convolution; another summation order inside the CPU conv kernels)
    perm2: float32, vertically flipped operands + input channels rotated by half
    f64  : the reference in float64 (weights, frames, activations)
    sepflip: float32, SepConv plugin only: the 51-tap separable op (forward and gradients) evaluated on spatially flipped
           operands with reversed tap order -- the same op, its 2601-term sums in the opposite order (the conv2d variants above
           leave the op's own rounding untouched; a build that changes the op's kernels perturbs exactly this)
    meanflip: float32, CAIN plugin only: sub_mean's per-channel mean taken over W first, then over H (the reference: H, then W) -- the
           same mean, rounded differently in its last bit; the frames entering the network move by one ulp (a build that computes
           the mean with its own kernel, csrc/submean.hip, perturbs exactly this)

and stores, per case and phase, the deviation of perm / perm2 / f64 from base in exactly the normalisation
the GPU parity tests use (loss: relative; preds: mean |.|; PSNR / SSIM: absolute; fingerprints: relative to
the abs-sum scale, max over tensors).  tests/test_system_gpu.py gates the HIP path at
max(north-star bound, K x the float32 self-spread) -- i.e. "as close to the reference as the reference is to
itself under another summation order".
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)

from oracle import gen_golden as G  # noqa: E402
from oracle import torch_ops as O  # noqa: E402
from meta_interpolation_amd import synthetic  # noqa: E402

CASES = ['voxelflow_metasgd_adamax_2step', 'sepconv_metasgd_adamax_2step', 'cain_lslr_adam_1step',
         'voxelflow_lslr_sgd_2step', 'voxelflow_script_metasgd_adam_1step', 'sepconv_lslr_sgd_2step',
         'superslomo_lslr_sgd_2step', 'c1_cain_lslr_sgd', 'sepconv_msl_learnable_2step', 'cain_l2f', 'rrin_lslr_sgd_2step']
VARIANTS = ['perm', 'perm2', 'f64', 'sepflip', 'meanflip']

_ORIG_CONV2D = torch.nn.functional.conv2d


def _conv2d_perm(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    x, w, b = input, weight, bias
    if groups != 1:
        return _ORIG_CONV2D(x, w, b, stride, padding, dilation, groups)
    ci = torch.arange(w.shape[1] - 1, -1, -1)
    y = _ORIG_CONV2D(x[:, ci].flip(3), w[:, ci].flip(3), b, stride, padding, dilation, groups)
    return y.flip(3)


def _conv2d_perm2(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    x, w, b = input, weight, bias
    if groups != 1:
        return _ORIG_CONV2D(x, w, b, stride, padding, dilation, groups)
    n = w.shape[1]
    ci = torch.roll(torch.arange(n), n // 2)
    y = _ORIG_CONV2D(x[:, ci].flip(2), w[:, ci].flip(2), b, stride, padding, dilation, groups)
    return y.flip(2)


class _SepconvAnyDtype(torch.autograd.Function):
    """The op oracle for float64 runs: the differentiable PyTorch statement (oracle/torch_ops.sepconv_torch)."""

    @staticmethod
    def apply(inp, v, h):
        return O.sepconv_torch(inp, v, h)


class _SepconvFlipped:
    """The float32 op oracle on flipped operands: out'(y', x') = out(Ho - 1 - y', Wo - 1 - x') with the taps walked backwards."""

    @staticmethod
    def apply(inp, v, h):
        return O.SepconvCPU.apply(inp.flip(2, 3), v.flip(1, 2, 3), h.flip(1, 2, 3)).flip(2, 3)


def _sub_mean_flipped(x):
    mean = x.mean(3, keepdim=True).mean(2, keepdim=True)
    return x - mean, mean


def run_variant(name, variant, phase):
    model, H, W, B, over = G.SYSTEM_CASES[name]
    args = G.reference_args(model=model, batch_size=B, **over)
    frames = synthetic.septuplet_batch(B, H, W, model=model)
    torch.manual_seed(0)
    torch.nn.functional.conv2d = {'perm': _conv2d_perm, 'perm2': _conv2d_perm2}.get(variant, _ORIG_CONV2D)
    try:
        system = G.build_reference_system(args, model)
        if getattr(args, 'attenuate', False):
            rs = np.random.RandomState(777)
            with torch.no_grad():
                system.gamma_mult.fill_(0.5)
                for p in system.attenuator.parameters():
                    p.copy_(torch.from_numpy(rs.uniform(-0.05, 0.05, size=tuple(p.shape)).astype(np.float32)))
        if variant == 'f64':
            system.double()
            if hasattr(system, 'mean'):
                system.mean, system.std = system.mean.double(), system.std.double()
            frames = [f.double() for f in frames]
            import utils as ref_utils          # PSNR / SSIM of the float64 outputs through the float32 metric code (its SSIM window is float32)
            if not hasattr(ref_utils, '_savfi_orig_calc_metrics'):
                ref_utils._savfi_orig_calc_metrics = ref_utils.calc_metrics
            ref_utils.calc_metrics = lambda a, b: ref_utils._savfi_orig_calc_metrics(a.float(), b.float())
            if model == 'sepconv':
                import sepconv.sepconv_op.sepconv as ref_op
                ref_op.FunctionSepconv = _SepconvAnyDtype
        if variant == 'sepflip' and model == 'sepconv':
            import sepconv.sepconv_op.sepconv as ref_op
            ref_op.FunctionSepconv = _SepconvFlipped
        if variant == 'meanflip' and model == 'cain':
            import cain.model as ref_cain
            ref_cain._savfi_orig_sub_mean = ref_cain.sub_mean
            ref_cain.sub_mean = _sub_mean_flipped
        rec = dict(n_live=[], grad_fp=[], weight_fp=[], outer_grad_fp={})
        G.observe(system, rec)
        if phase == 'train':
            losses, preds, metrics = system.run_train_iter(data_batch=[f.clone() for f in frames], epoch=0, do_evaluation=True)
        else:
            losses, preds, metrics = system.run_validation_iter(data_batch=[f.clone() for f in frames])
    finally:
        torch.nn.functional.conv2d = _ORIG_CONV2D
        import utils as ref_utils
        if hasattr(ref_utils, '_savfi_orig_calc_metrics'):
            ref_utils.calc_metrics = ref_utils._savfi_orig_calc_metrics
        if model == 'sepconv':
            import sepconv.sepconv_op.sepconv as ref_op
            ref_op.FunctionSepconv = O.SepconvCPU
        if model == 'cain':
            import cain.model as ref_cain
            if hasattr(ref_cain, '_savfi_orig_sub_mean'):
                ref_cain.sub_mean = ref_cain._savfi_orig_sub_mean
    return dict(loss=float(losses['loss'].item()), preds=torch.stack([p.squeeze(0) for p in preds]).double().numpy(),
                psnr=float(metrics['psnr'].avg), ssim=float(metrics['ssim'].avg), rec=rec)


def fp_dev(a, b):
    """max over tensors of the fingerprint deviation, relative to the abs-sum scale (tests.helpers.assert_fp_close)."""
    worst = 0.0
    for k in a:
        if k not in b:
            continue
        scale = max(abs(a[k][1]), 1e-12)
        worst = max(worst, abs(a[k][0] - b[k][0]) / scale, abs(a[k][1] - b[k][1]) / scale)
    return worst


def deviations(base, other):
    d = dict(loss=abs(other['loss'] - base['loss']) / max(abs(base['loss']), 1e-30),
             l1=float(np.abs(other['preds'] - base['preds']).mean()),
             psnr=abs(other['psnr'] - base['psnr']), ssim=abs(other['ssim'] - base['ssim']))
    rb, ro = base['rec'], other['rec']
    d['n_live_equal'] = float(rb['n_live'] == ro['n_live'])
    steps = min(len(rb['weight_fp']), len(ro['weight_fp']))
    d['w'] = max([fp_dev(rb['weight_fp'][i], ro['weight_fp'][i]) for i in range(steps)] or [0.0])
    d['g0'] = fp_dev(rb['grad_fp'][0], ro['grad_fp'][0]) if steps else 0.0
    d['g'] = max([fp_dev(rb['grad_fp'][i], ro['grad_fp'][i]) for i in range(steps)] or [0.0])
    d['outer'] = fp_dev(rb['outer_grad_fp'], ro['outer_grad_fp']) if rb['outer_grad_fp'] else 0.0
    return d


QUANT = ['loss', 'l1', 'psnr', 'ssim', 'w', 'g0', 'g', 'outer', 'n_live_equal']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', nargs='*', default=None)
    opts = ap.parse_args()
    torch.set_num_threads(8)
    G.install_shims()
    path = os.path.join(G.GOLD, 'sensitivity.npz')
    out = dict(np.load(path)) if os.path.exists(path) else {}
    out['variants'] = np.array(VARIANTS)
    out['quantities'] = np.array(QUANT)
    for name in (opts.only or CASES):
        fx = np.load(os.path.join(G.GOLD, 'system_%s.npz' % name))
        for phase in ('train', 'val'):
            base = run_variant(name, 'base', phase)
            # the float32 run must BE the committed fixture
            assert abs(base['loss'] - float(fx[phase + '_loss'])) <= 1e-7 * abs(base['loss']), (name, phase, base['loss'])
            assert np.abs(base['preds'] - fx[phase + '_preds']).max() <= 1e-6, (name, phase)
            table = np.zeros((len(VARIANTS), len(QUANT)))
            for vi, variant in enumerate(VARIANTS):
                d = deviations(base, run_variant(name, variant, phase))
                table[vi] = [d[q] for q in QUANT]
                print('  %-36s %-5s %-5s ' % (name, phase, variant) + ' '.join('%s=%.2e' % (q, d[q]) for q in QUANT), flush=True)
            out['%s/%s' % (name, phase)] = table
        np.savez_compressed(path, **out)


if __name__ == '__main__':
    main()
