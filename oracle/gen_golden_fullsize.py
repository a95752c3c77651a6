"""oracle/gen_golden_fullsize.py -- TEST INFRASTRUCTURE.  Full-size fixtures for the BASELINE.json configs, by IMPORTING THE REFERENCE.

    python oracle/gen_golden_fullsize.py [--only name ...]     # writes tests/golden/full_<name>.npz

The 64x64 fixtures of gen_golden.py exercise every code path but not the kernel mix of the benchmark shapes (at 64x64 most
layers sit below the Winograd / weight-gradient thresholds of the HIP path).  These cases run the imported reference -- same
shims as gen_golden.py, no reference file modified -- at the sizes BASELINE.json quotes, one task each (tasks are independent):

    c2_sepconv_256x448_s5        SepConv, LSLR + SGD, 5 inner steps, 256x448             (config C2 / C4's per-task work)
    c3_voxelflow_256x256_s5      VoxelFlow, Meta-SGD + Adamax, 5 inner steps, 256x256    (config C3; + the reference's own
                                 spread under another conv summation order / float64, like gen_sensitivity.py)
    c3sgd_voxelflow_256x256_s5   the same with the smooth LSLR + SGD rule
    c2b4_sepconv_256x448_s5      config C2 as benchmarked: the reference's run_train_iter over a meta-batch of FOUR tasks (seeds
                                 1234+t) -- what the product adapts in lockstep (T=4) / from graphs on task streams
    c4_sepconv_msl_256x448_s5    one GPU's share of config C4: 4 tasks, MAML++ multi-step loss (a weighted target pass after
                                 every inner step) + learnable per-layer per-step learning rates
    c2script_sepconv_256x448_b3_s3   the configuration the reference's own scripts/run_sepconv.sh:6-17 trains: Adamax + Meta-SGD (element-wise
                                 learnable learning rates, inner_loop_optimizers.py:385-425), 3 inner steps, meta-batch 3, inner_lr 1e-5
    c4b32_sepconv_msl_256x448_s5 config C4 at its stated meta-batch of 32 (meta_learning_system.py:338,366).  The imported reference keeps every
                                 task's five target-pass graphs until its one backward: 65 GB at 32 tasks, more than this container has
                                 (62 GB: the kernel's OOM killer ended the attempt).  The 32-task iteration is therefore assembled from FOUR
                                 reference run_train_iter calls over tasks 8 g .. 8 g + 7 at the same theta -- tasks are independent
                                 (first-order) and the reference's loss / outer gradient are task MEANS, so the 32-task values are the means
                                 of the four groups' (linear; exact up to one fp32 rounding of the averaging).  Stored: mean loss / PSNR /
                                 SSIM, the frames of tasks 0 and 31, outer-gradient fingerprints (sum and first elements averaged; the
                                 abs-sum scale is the mean of the groups' abs-sums, an upper bound of the true one)
    c5_cain_l2f_720p             CAIN + L2F attenuation, 1 inner step, 1280x720, run_train_iter   (config C5)
    c5eval_cain_l2f_720p         the reference's ExperimentBuilder.evaluation_iteration (experiment_builder.py:93-148) on the
                                 same clip: 720x1280 > 5e5 pixels, so two 720x640 halves are adapted separately and stitched

Stored: loss, loss parts, PSNR / SSIM, per-step gradient / fast-weight fingerprints, outer-gradient fingerprints, and the
predicted frame (float32 up to 256x448; the 720p frames every second pixel as uint16 over [-0.25, 1.25], 1.1e-5 resolution,
to keep the fixture at a few MB).
"""
import argparse
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)

from oracle import gen_golden as G  # noqa: E402
from oracle import gen_sensitivity as S  # noqa: E402
from meta_interpolation_amd import synthetic  # noqa: E402

CASES = {
    'c2_sepconv_256x448_s5': ('sepconv', 256, 448, dict(optimizer='SGD', inner_lr=1e-3, loss='1*L1',
                                                         number_of_training_steps_per_iter=5,
                                                         number_of_evaluation_steps_per_iter=5)),
    'c2b4_sepconv_256x448_s5': ('sepconv', 256, 448, dict(optimizer='SGD', inner_lr=1e-3, loss='1*L1',
                                                           number_of_training_steps_per_iter=5,
                                                           number_of_evaluation_steps_per_iter=5)),
    'c4_sepconv_msl_256x448_s5': ('sepconv', 256, 448, dict(optimizer='SGD', inner_lr=1e-3, loss='1*L1',
                                                             number_of_training_steps_per_iter=5,
                                                             number_of_evaluation_steps_per_iter=5,
                                                             use_multi_step_loss_optimization=True,
                                                             multi_step_loss_num_epochs=10,
                                                             learnable_per_layer_per_step_inner_loop_learning_rate=True)),
    'c2script_sepconv_256x448_b3_s3': ('sepconv', 256, 448, dict(optimizer='Adamax', inner_lr=1e-5, metasgd=True, loss='1*L1',
                                                                 number_of_training_steps_per_iter=3,
                                                                 number_of_evaluation_steps_per_iter=3)),
    'c4b32_sepconv_msl_256x448_s5': ('sepconv', 256, 448, dict(optimizer='SGD', inner_lr=1e-3, loss='1*L1',
                                                                number_of_training_steps_per_iter=5,
                                                                number_of_evaluation_steps_per_iter=5,
                                                                use_multi_step_loss_optimization=True,
                                                                multi_step_loss_num_epochs=10,
                                                                learnable_per_layer_per_step_inner_loop_learning_rate=True)),
    'c3_voxelflow_256x256_s5': ('voxelflow', 256, 256, dict(optimizer='Adamax', inner_lr=1e-5, metasgd=True, loss='1*MSE',
                                                             number_of_training_steps_per_iter=5,
                                                             number_of_evaluation_steps_per_iter=5)),
    'c3sgd_voxelflow_256x256_s5': ('voxelflow', 256, 256, dict(optimizer='SGD', inner_lr=1e-3, loss='1*MSE',
                                                                number_of_training_steps_per_iter=5,
                                                                number_of_evaluation_steps_per_iter=5)),
    # config C3 as benchmarked AND testable: the same rule / sizes / 8-task meta-batch with VoxelFlow's weights at the model's own
    # initialisation scale (synthetic.py recipe 'smooth': sub-pixel, smooth flows) -- the reference reproduces itself on it
    'c3s_voxelflow_256x256_b8_s5': ('voxelflow', 256, 256, dict(optimizer='Adamax', inner_lr=1e-5, metasgd=True, loss='1*MSE',
                                                                 number_of_training_steps_per_iter=5,
                                                                 number_of_evaluation_steps_per_iter=5)),
    'c5_cain_l2f_720p': ('cain', 720, 1280, dict(optimizer='SGD', inner_lr=1e-3, attenuate=True, loss='1*L1')),
    'c5eval_cain_l2f_720p': ('cain', 720, 1280, dict(optimizer='SGD', inner_lr=1e-3, attenuate=True, loss='1*L1')),
}
TASKS = {'c2b4_sepconv_256x448_s5': 4, 'c4_sepconv_msl_256x448_s5': 4, 'c3s_voxelflow_256x256_b8_s5': 8, 'c2script_sepconv_256x448_b3_s3': 3,
         'c4b32_sepconv_msl_256x448_s5': 32}
KEEP_TASKS = {'c4b32_sepconv_msl_256x448_s5': (0, 31)}     # which tasks' predicted frames a fixture stores (default: all)
RECIPE = {'c3s_voxelflow_256x256_b8_s5': 'smooth'}      # seeded-weights recipe (synthetic.py); stored in the fixture's args as weight_recipe      # meta-batch size (default 1)
SPREAD_FOR = {'c3_voxelflow_256x256_s5', 'c3sgd_voxelflow_256x256_s5', 'c2_sepconv_256x448_s5', 'c3s_voxelflow_256x256_b8_s5',
              'c2script_sepconv_256x448_b3_s3'}
Q_LO, Q_HI = -0.25, 1.25


def pack_pred(pred, compact=False):
    """[3,H,W] float tensor -> dict of arrays (see the module docstring)."""
    p = pred.detach().float()
    if p.shape[-2] * p.shape[-1] <= 256 * 448 and not compact:
        return {'pred': p.numpy()}
    # (seeded weights: a frame may leave [-0.25, 1.25] -- the range widens to hold it, the resolution (range / 65535) with it; the
    # tests add half a step to their gate)
    lo, hi = min(Q_LO, float(p.min())), max(Q_HI, float(p.max()))
    q = ((p[:, ::2, ::2] - lo) / (hi - lo) * 65535.0).round().numpy().astype(np.uint16)
    return {'pred_u16_stride2': q, 'pred_q_range': np.array([lo, hi])}


def seed_attenuator(system):
    rs = np.random.RandomState(777)
    with torch.no_grad():
        system.gamma_mult.fill_(0.5)
        for p in system.attenuator.parameters():
            p.copy_(torch.from_numpy(rs.uniform(-0.05, 0.05, size=tuple(p.shape)).astype(np.float32)))


GROUPS = {'c4b32_sepconv_msl_256x448_s5': 4}       # a meta-batch assembled from this many reference calls (see the module docstring)


def run_train(name, variant='base', group=None):
    model, H, W, over = CASES[name]
    B = TASKS.get(name, 1)
    first = 0
    if group is not None:
        B = B // GROUPS[name]
        first = group * B
    args = G.reference_args(model=model, batch_size=B, **over)
    frames = synthetic.septuplet_batch(B, H, W, model=model, first_task=first)
    torch.manual_seed(0)
    torch.nn.functional.conv2d = {'perm': S._conv2d_perm, 'perm2': S._conv2d_perm2}.get(variant, S._ORIG_CONV2D)
    try:
        system = G.build_reference_system(args, model, recipe=RECIPE.get(name))
        if getattr(args, 'attenuate', False):
            seed_attenuator(system)
        if variant == 'f64':
            system.double()
            if hasattr(system, 'mean'):
                system.mean, system.std = system.mean.double(), system.std.double()
            frames = [f.double() for f in frames]
            import utils as ref_utils
            if not hasattr(ref_utils, '_savfi_orig_calc_metrics'):
                ref_utils._savfi_orig_calc_metrics = ref_utils.calc_metrics
            ref_utils.calc_metrics = lambda a, b: ref_utils._savfi_orig_calc_metrics(a.float(), b.float())
            if model == 'sepconv':
                import sepconv.sepconv_op.sepconv as ref_op
                ref_op.FunctionSepconv = S._SepconvAnyDtype
        rec = dict(n_live=[], grad_fp=[], weight_fp=[], outer_grad_fp={})
        G.observe(system, rec)
        losses, preds, metrics = system.run_train_iter(data_batch=[f.clone() for f in frames], epoch=0, do_evaluation=True)
    finally:
        torch.nn.functional.conv2d = S._ORIG_CONV2D
        import utils as ref_utils
        if hasattr(ref_utils, '_savfi_orig_calc_metrics'):
            ref_utils.calc_metrics = ref_utils._savfi_orig_calc_metrics
        if model == 'sepconv':
            import sepconv.sepconv_op.sepconv as ref_op
            from oracle import torch_ops as O
            ref_op.FunctionSepconv = O.SepconvCPU
    return dict(loss=float(losses['loss'].item()), parts={k: float(v) for k, v in losses.items()
                                                          if k != 'loss' and not k.startswith('loss_importance')},
                preds=torch.stack([p.squeeze(0) for p in preds]).double().numpy(),
                psnr=float(metrics['psnr'].avg), ssim=float(metrics['ssim'].avg), rec=rec)


def run_train_grouped(name):
    """The meta-batch as GROUPS[name] reference calls at the same theta; returns run_train's dict for the whole meta-batch."""
    n = GROUPS[name]
    parts = []
    for g in range(n):
        parts.append(run_train(name, group=g))
        print('  %-30s group %d / %d: loss=%.8f' % (name, g + 1, n, parts[-1]['loss']), flush=True)
    mean = lambda xs: float(np.mean(xs))
    rec = dict(n_live=parts[0]['rec']['n_live'], grad_fp=[], weight_fp=[], outer_grad_fp={})
    for k in parts[0]['rec']['outer_grad_fp']:
        rec['outer_grad_fp'][k] = np.mean(np.stack([p['rec']['outer_grad_fp'][k] for p in parts]), axis=0)
    return dict(loss=mean([p['loss'] for p in parts]), parts={k: mean([p['parts'][k] for p in parts]) for k in parts[0]['parts']},
                preds=np.concatenate([p['preds'] for p in parts]), psnr=mean([p['psnr'] for p in parts]),
                ssim=mean([p['ssim'] for p in parts]), rec=rec)


def gen_train_case(name):
    model, H, W, over = CASES[name]
    t0 = time.time()
    base = run_train_grouped(name) if name in GROUPS else run_train(name)
    stored = dict(over, weight_recipe=RECIPE[name]) if name in RECIPE else over
    out = {'model': np.array(model), 'H': H, 'W': W, 'B': TASKS.get(name, 1), 'args': np.array(repr(sorted(stored.items()))),
           'train_loss': np.float64(base['loss']), 'train_psnr': np.float64(base['psnr']), 'train_ssim': np.float64(base['ssim']),
           'train_n_live': np.array(base['rec']['n_live'])}
    for k, v in base['parts'].items():
        out['train_part_' + k] = np.float64(v)
    out.update({'train_' + k: v for k, v in pack_pred(torch.from_numpy(base['preds'][0])).items()})
    for t in range(1, TASKS.get(name, 1)):       # further tasks of the meta-batch: the compact form (keeps the fixture small)
        if name in KEEP_TASKS and t not in KEEP_TASKS[name]:
            continue
        out.update({'train_task%d_%s' % (t, k): v for k, v in pack_pred(torch.from_numpy(base['preds'][t]), compact=True).items()})
    G.pack_fp('train_grad_fp', base['rec']['grad_fp'], out)
    G.pack_fp('train_weight_fp', base['rec']['weight_fp'], out)
    G.pack_fp('outer_grad_fp', [base['rec']['outer_grad_fp']], out)
    print('  %-30s loss=%.8f psnr=%.4f n_live=%s  (%.0f s)' % (name, base['loss'], base['psnr'], base['rec']['n_live'],
                                                                time.time() - t0), flush=True)
    if name in SPREAD_FOR:
        table = np.zeros((len(S.VARIANTS), len(S.QUANT)))
        for vi, variant in enumerate(S.VARIANTS):
            d = S.deviations(base, run_train(name, variant))
            table[vi] = [d[q] for q in S.QUANT]
            print('  %-30s %-5s ' % (name, variant) + ' '.join('%s=%.2e' % (q, d[q]) for q in S.QUANT), flush=True)
        out['spread'] = table
        out['spread_variants'] = np.array(S.VARIANTS)
        out['spread_quantities'] = np.array(S.QUANT)
    np.savez_compressed(os.path.join(G.GOLD, 'full_%s.npz' % name), **out)


def gen_eval_case(name):
    """The reference's own ExperimentBuilder.evaluation_iteration, called unbound with a stand-in `self` (its constructor
    needs tensorboard and a data provider; the method itself only touches args, model, epoch and the summary string)."""
    import experiment_builder as ref_eb
    model, H, W, over = CASES[name]
    args = G.reference_args(model=model, batch_size=1, **over)
    torch.Tensor.cuda = lambda self, *a, **k: self            # target = images[3][0].detach().cuda()  (:133)
    system = G.build_reference_system(args, model)
    seed_attenuator(system)
    frames = synthetic.septuplet_batch(1, H, W, model=model)
    calls = []
    orig = system.run_validation_iter
    system.run_validation_iter = lambda data_batch: (calls.append(tuple(data_batch[0].shape)), orig(data_batch=data_batch))[1]
    stub = types.SimpleNamespace(args=args, model=system, epoch=0,
                                 build_loss_summary_string=lambda losses, metrics: '')
    pbar = types.SimpleNamespace(update=lambda n: None, set_description=lambda s: None)
    t0 = time.time()
    losses, outputs, metrics = ref_eb.ExperimentBuilder.evaluation_iteration(
        stub, val_sample=([f.clone() for f in frames], {}), total_losses={}, pbar_val=pbar, phase='val')
    out = {'model': np.array(model), 'H': H, 'W': W, 'B': 1, 'args': np.array(repr(sorted(over.items()))),
           'val_loss': np.float64(float(losses['loss'])), 'val_psnr': np.float64(metrics['psnr'].avg),
           'val_ssim': np.float64(float(metrics['ssim'].avg)), 'half_shapes': np.array(calls)}
    for k, v in losses.items():
        if k != 'loss' and not k.startswith('loss_importance'):
            out['val_part_' + k] = np.float64(float(v))
    out.update({'val_' + k: v for k, v in pack_pred(outputs[0].squeeze(0)).items()})
    print('  %-30s loss=%.8f psnr=%.4f halves=%s  (%.0f s)' % (name, out['val_loss'], out['val_psnr'], calls, time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(G.GOLD, 'full_%s.npz' % name), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', nargs='*', default=None)
    opts = ap.parse_args()
    torch.set_num_threads(int(os.environ.get('ORACLE_THREADS', 8)))
    G.install_shims()
    for name in (opts.only or list(CASES)):
        print('[golden full-size]', name, flush=True)
        (gen_eval_case if name.startswith('c5eval') else gen_train_case)(name)


if __name__ == '__main__':
    main()
