"""-m gpu: the HIP product path (SceneAdaptiveInterpolation + plugins + fused rules, every custom op
through libsavfi_hip.so) against (a) the golden fixtures generated from the imported reference and
(b) the CPU oracle, on identical seeded weights and septuplets.

Gates (SURVEY.md 8d "Parity gate"): output frames <= 1e-4 mean abs (pixel L1, [0,1] scale) and
<= 1e-3 dB PSNR; losses rel 1e-5; fast-weight fingerprints rel 1e-5; gradient fingerprints rel 1e-3 of
their abs-sum -- or K x the reference's own measured spread where that is larger (see `tolerances`).
"""
import numpy as np
import pytest
import torch

from meta_interpolation_amd import _hip, hip_ops, synthetic
from meta_interpolation_amd.config import default_args
from meta_interpolation_amd.inner_loop_optimizers import LSLRGradientDescentLearningRule, MetaSGDLearningRule
from meta_interpolation_amd.meta_learning_system import SceneAdaptiveInterpolation
from oracle import rules as orules
from tests.helpers import fp as helpers_fp
from tests.helpers import FP_ATOL, assert_fp_close, build_plugin, build_system, fp, golden, observe, parse_case_args

pytestmark = pytest.mark.gpu
DEV = "cuda"

SYSTEM = ['c1_cain_lslr_sgd', 'cain_l2f', 'cain_lslr_adam_1step', 'sepconv_lslr_sgd_2step',
          'sepconv_metasgd_adamax_2step', 'sepconv_msl_learnable_2step', 'voxelflow_metasgd_adamax_2step',
          'voxelflow_lslr_sgd_2step', 'voxelflow_script_metasgd_adam_1step', 'rrin_lslr_sgd_2step',
          'superslomo_lslr_sgd_2step',
          # 128 x 128 twins (deepest maps 4 x 4): the OUTER-gradient fingerprints at the plain 1e-3 gate
          'sepconv_msl_learnable_2step_128', 'superslomo_lslr_sgd_2step_128']

# Gates = max(contract bound, K x the REFERENCE's own spread).
#
# Contract (north star / SURVEY.md 8d): pixel L1 <= 1e-4, PSNR <= 1e-3 dB, loss rel 1e-5, fast-weight fingerprints rel
# 1e-5 (+ SSIM 1e-4 and gradient fingerprints 1e-3 of their abs-sum, which are this suite's own additions).
#
# tests/golden/sensitivity.npz (oracle/gen_sensitivity.py) holds, per case and phase, how far the imported reference moves
# away from ITSELF when its float32 convolutions are summed in another order (two operand permutations) and when it runs in
# float64.  SGD-type rules: that spread is far below the contract and the contract bound is the gate.  Adam / Adamax-type
# rules step +-lr*c per element whatever |g| (g/(|g|+1e-8)): elements whose gradient is below conv rounding noise flip, and
# VoxelFlow's flow-to-pixel map amplifies that over further steps -- the reference's 2-step VoxelFlow+Meta-SGD+Adamax run
# differs from its own float64 run by 1.8e-3 pixel L1 / 3.3e-3 dB.  No implementation, the reference's CUDA path included,
# can be closer to the CPU reference than the CPU reference is to itself, so for those quantities the gate is K x the
# measured spread (K = 3 for the contract quantities, K = 5 for gradient fingerprints: MIOpen / Winograd kernels differ from
# the CPU's direct convolution by algorithm, not only by summation order).  The step-level checks that do not compound --
# gradients at theta, "fused rule == oracle rule on identical inputs", and the teacher-forced test further down -- stay
# at the contract bounds for every case.
CONTRACT = dict(loss=1e-5, l1=1e-4, psnr=1e-3, ssim=1e-4, w=1e-5, g=1e-3, outer=1e-3)
K_SPREAD = dict(loss=3, l1=3, psnr=3, ssim=3, w=3, g=5, outer=5)
_SENS = golden("sensitivity")
_SENS_COL = {q: i for i, q in enumerate(_SENS['quantities'].tolist())}


def tolerances(name, phase='train'):
    tol = dict(CONTRACT)
    key = '%s/%s' % (name, phase)
    if key in _SENS.files:
        table = _SENS[key]                              # [variant, quantity]
        for q in tol:
            tol[q] = max(tol[q], K_SPREAD[q] * float(table[:, _SENS_COL[q]].max()))
    # Learning-rate gradients of the SepConv plugin on 64 x 64 frames: its Conv4 / Conv5 / Deconv layers have 8 x 8 .. 2 x 2 maps there, one
    # ReLU unit switching under another summation order moves such a layer's learning-rate gradient by ~1e-3, and the fingerprints take a
    # few discrete values (1.3e-4 / 7e-4 / 1.2e-3 / 4.1e-3) that the SAME build hits in different processes -- with the round-2 kernels
    # as with this round's more accurate ones (profiles/r03_lr_gradient_modes_sepconv64.txt).  The 1e-3 gate passed or failed by the
    # draw; 5e-3 covers the observed modes.  Losses, predictions, weights and per-step gradients keep their gates.
    if name.startswith('sepconv_') and phase == 'train' and int(golden('system_' + name)['H']) <= 64:      # the small-map argument only
        tol['outer'] = max(tol['outer'], 5e-3)
    # Super SloMo's outer-gradient fingerprints: 6.3e-4 in most processes, 4.9e-3 (net.flowComp.conv2.bias) in about one process
    # of three -- the same value in all four fixture tests of that process, another process of the same build back at 6.3e-4
    # (profiles/r03_superslomo_outer_gradient_modes.txt: three full runs of this suite on one box).  The fixture's deep layers
    # (<= 8 x 8 maps) run on MIOpen, whose find step picks solvers per process; a LeakyReLU unit of such a map switching is ~5e-3 of
    # that bias's learning-rate gradient: the SepConv case above.  Rounds 1-2 carried 1e-2 here (then for the 7x7 / 5x5 stages on
    # MIOpen's implicit-GEMM kernels, 3.1e-3 off); the gate removed early this round made the suite fail one run in three.
    if name.startswith('superslomo_') and phase == 'train' and int(golden('system_' + name)['H']) <= 64:
        tol['outer'] = max(tol['outer'], 1e-2)
    return tol



def knee_allowance(g, phase, step, key, tol_g):
    """Absolute allowance on an updated weight's fingerprint sums under an Adam-type inner rule, derived from the GRADIENT gate.
    Those rules move an element by lr * g / (|g| + 1e-8) (first step: m-hat = g, sqrt(v-hat) = |g|; Adamax: u = |g|): for |g| >> 1e-8
    the step is +-lr whatever the rounding of g, but an element whose reference gradient is exactly 0 (a dead ReLU unit) or ~1e-9
    sits on the knee, where a gradient error d moves the step by lr * d / (d + 1e-8).  The gradient gate admits d = tol_g x (the
    tensor's gradient abs-sum); concentrated in ONE element that is the allowance returned here (several knee elements would add
    up to more: not granted).  Seen as 2.1e-6 = 0.02 lr on a 12-element channel-attention bias (abs-sum 0.116, gate 1e-5 relative)
    of cain_lslr_adam_1step when sub_mean moved to csrc/submean.hip and the frames entering the network moved by one ulp; the
    fixture shows an exactly-zero element in that tensor's reference gradient.  SGD-type rules are smooth: 0."""
    args = parse_case_args(g)
    if args.get('optimizer') not in ('Adam', 'Adamax'):
        return 0.0
    keys = list(g['%s_grad_fp_%d_keys' % (phase, step)])
    if key not in keys:
        return 0.0
    d = tol_g * abs(float(g['%s_grad_fp_%d' % (phase, step)][keys.index(key)][1]))
    return float(args['inner_lr']) * d / (d + 1e-8)


TOL = {name: tolerances(name) for name in SYSTEM}


def run_case(name, phase, fuse=1, check_rule=False):
    g = golden("system_" + name)
    model = str(g['model'])
    # the reference's sequential task loop: per-step fingerprints and the 94 -> 54 live-tensor counts are defined on it
    # (tasks in lockstep, the product default for multi-task meta-batches, have their own fixture tests below)
    system = build_system(model, dict(parse_case_args(g), task_batch=0), fuse=fuse)
    rec = observe(system, check_rule=check_rule)
    frames = synthetic.septuplet_batch(int(g['B']), int(g['H']), int(g['W']), model=model)
    if phase == 'train':
        losses, preds, metrics = system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=True)
    else:
        losses, preds, metrics = system.run_validation_iter(data_batch=frames)
    torch.cuda.synchronize()
    return g, losses, preds, metrics, rec


@pytest.mark.parametrize("name", SYSTEM)
@pytest.mark.parametrize("phase", ["train", "val"])
def test_iteration_matches_reference_fixture(name, phase):
    tol = tolerances(name, phase)
    g, losses, preds, metrics, rec = run_case(name, phase, check_rule=True)
    # every fused update in the loop == the oracle's rule applied to the same weights / grads / lrs
    assert max(rec['rule_err']) <= 1e-6, rec['rule_err']
    assert list(g[phase + '_n_live']) == rec['n_live']                           # fact 6: 94 -> 54, 23 -> 9
    # step 0 (before any update can amplify rounding differences): gradients at theta
    for k, row in zip(list(g['%s_grad_fp_0_keys' % phase]), g['%s_grad_fp_0' % phase]):
        assert_fp_close(rec['grad_fp'][0][k], row, 1e-3, (name, 'g0', k))
    want_loss = float(g[phase + '_loss'])
    assert abs(losses['loss'].item() - want_loss) <= tol['loss'] * abs(want_loss)
    got = torch.stack([p.squeeze(0) for p in preds]).cpu().numpy()
    assert np.abs(got - g[phase + '_preds']).mean() < tol['l1']                  # pixel L1 gate
    assert abs(metrics['psnr'].avg - float(g[phase + '_psnr'])) < tol['psnr']    # dB gate
    assert abs(float(metrics['ssim'].avg) - float(g[phase + '_ssim'])) < tol['ssim']
    for i, d in enumerate(rec['weight_fp']):
        keys = list(g['%s_weight_fp_%d_keys' % (phase, i)])
        assert sorted(d) == keys
        for k, row in zip(keys, g['%s_weight_fp_%d' % (phase, i)]):
            assert_fp_close(d[k], row, tol['w'], (name, 'w', i, k), extra_abs=knee_allowance(g, phase, i, k, tol['g']))
    for i, d in enumerate(rec['grad_fp']):
        for k, row in zip(list(g['%s_grad_fp_%d_keys' % (phase, i)]), g['%s_grad_fp_%d' % (phase, i)]):
            assert_fp_close(d[k], row, tol['g'], (name, 'g', i, k))
    if phase == 'train':
        rows = dict(zip(list(g['outer_grad_fp_0_keys']), g['outer_grad_fp_0']))
        assert set(rec['outer_grad_fp']) == set(rows)
        for k, row in rows.items():
            assert_fp_close(rec['outer_grad_fp'][k], row, tol['outer'], (name, 'outer', k))


@pytest.mark.parametrize("name", [n for n in SYSTEM if n not in ('rrin_lslr_sgd_2step', 'superslomo_lslr_sgd_2step')])
def test_iteration_matches_reference_fixture_under_default_miopen_solvers(name):
    """The same fixtures with MIOpen's DEFAULT (non-deterministic, atomics-based) fp32 solvers -- the solver set bench.py and
    the product run with (conftest.py pins the deterministic ones for every other test).  Contract quantities only:
    gradient fingerprints wander at the 1e-4 level run to run under atomic accumulation (profiles/r01_determinism_survey.txt)."""
    tol = tolerances(name, 'train')
    prev = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = False
    try:
        g, losses, preds, metrics, rec = run_case(name, 'train', check_rule=False)
    finally:
        torch.backends.cudnn.deterministic = prev
    assert list(g['train_n_live']) == rec['n_live']
    want_loss = float(g['train_loss'])
    assert abs(losses['loss'].item() - want_loss) <= tol['loss'] * abs(want_loss)
    got = torch.stack([p.squeeze(0) for p in preds]).cpu().numpy()
    assert np.abs(got - g['train_preds']).mean() < tol['l1']
    assert abs(metrics['psnr'].avg - float(g['train_psnr'])) < tol['psnr']
    assert abs(float(metrics['ssim'].avg) - float(g['train_ssim'])) < tol['ssim']
    for i, d in enumerate(rec['weight_fp']):
        for k, row in zip(list(g['train_weight_fp_%d_keys' % i]), g['train_weight_fp_%d' % i]):
            assert_fp_close(d[k], row, tol['w'], (name, 'w', i, k), extra_abs=knee_allowance(g, 'train', i, k, tol['g']))


# ---------------------------------------------------------------------------------------------
# Teacher-forced steps: the cases whose end-of-iteration gates are wider than the contract (sign-like inner rules).
# Every inner step of the HIP path starts from the ORACLE's weights W_t (computed live on the CPU; the oracle is pinned to
# the reference fixtures at 1e-5 by tests/test_oracle_golden.py) and must land on the oracle's W_{t+1}; the target pass
# with the oracle's adapted weights must give the oracle's frame.  Nothing compounds here, so the contract bounds apply:
# fast-weight fingerprints 1e-5, pixel L1 1e-4, PSNR 1e-3 dB -- plus a count of the elements that stepped the other way.
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ['voxelflow_metasgd_adamax_2step', 'sepconv_metasgd_adamax_2step', 'cain_lslr_adam_1step',
                                  'voxelflow_script_metasgd_adam_1step', 'voxelflow_lslr_sgd_2step'])
def test_teacher_forced_steps_meet_the_contract(name):
    from oracle import meta as ometa, models as omodels
    from tests.helpers import oracle_base
    g = golden("system_" + name)
    model, over = str(g['model']), parse_case_args(g)
    B, H, W = int(g['B']), int(g['H']), int(g['W'])
    S = int(over.get('number_of_training_steps_per_iter', 1))
    kind = 'metasgd' if over.get('metasgd') else 'lslr'
    opt, loss_kind = over['optimizer'], over['loss'].split('*')[1]
    system = build_system(model, over)
    system._set_pass_flags(False)
    frames = synthetic.septuplet_batch(B, H, W, model=model)
    frames_gpu = [f.to(DEV) for f in frames]
    base = oracle_base(model)
    names = ometa.inner_param_names([(n, p) for n, p in base.items() if p.is_floating_point()])
    lrs = orules.init_lrs(kind, {n: base[n] for n in names}, over['inner_lr'], num_steps=S)
    crit, fwd = ometa.criterion(loss_kind), omodels.FORWARD[model]
    torch.set_num_threads(16)
    flipped = total = 0
    for t in range(B):
        fast_o = {n: base[n] for n in names}
        st = orules.RuleState()
        system.inner_loop_optimizer.initialize_state()
        for step in range(S):
            sl = sum(crit(fwd(frames[i0][t][None], frames[i2][t][None], base, fast_o), frames[i1][t][None])
                     for i0, i1, i2 in ometa.SUPPORT)
            go = torch.autograd.grad(sl, list(fast_o.values()), allow_unused=True)
            next_o = orules.update_params(kind, opt, fast_o, dict(zip(fast_o.keys(), go)), lrs, step, st)
            Wg = {k: v.detach().to(DEV).requires_grad_() for k, v in fast_o.items()}
            loss_g = system._support_loss(frames_gpu, t, Wg, step)
            next_g = system.apply_inner_loop_update(loss_g, Wg, False, step)
            assert set(next_g) <= set(next_o) and len(next_g) > 0
            grads_o = dict(zip(fast_o.keys(), go))
            for k, v in next_g.items():
                want = next_o[k].detach()
                knee = 0.0
                if opt in ('Adam', 'Adamax') and grads_o.get(k) is not None:
                    # the knee of g / (|g| + 1e-8): see knee_allowance (here from the oracle's own gradient of this step)
                    d = CONTRACT['g'] * float(grads_o[k].detach().abs().sum())
                    knee = float(over['inner_lr']) * d / (d + 1e-8)
                assert_fp_close(fp(v), fp(want), CONTRACT['w'], (name, 'teacher-forced w', t, step, k), extra_abs=knee)
                # an element that moved the other way: more than half a full step (the largest move in its tensor) from
                # where the oracle put it.  (Measured against the tensor's step size, not the element's own move: elements with
                # an exactly-zero oracle gradient -- dead ReLU channels -- do not move at all.)
                step_size = (want - fast_o[k].detach()).abs().max().item()
                flipped += int(((v.detach().cpu() - want).abs() > 0.5 * step_size + 1e-12).sum())
                total += want.numel()
            fast_o = {k: v.detach().requires_grad_() for k, v in next_o.items()}
        with torch.no_grad():
            i0, i1, i2 = ometa.TARGET
            pred_o = fwd(frames[i0][t][None], frames[i2][t][None], base, fast_o)
            Wg = {k: v.detach().to(DEV) for k, v in fast_o.items()}
            _, pred_g = system._target_pass(frames_gpu, t, Wg, S)
            a = system._to_unit_range(pred_g.squeeze(0)).cpu()
            b = system._to_unit_range(pred_o.squeeze(0).to(DEV)).cpu()
            tgt = system._to_unit_range(frames_gpu[i1][t]).cpu()
        assert (a - b).abs().mean().item() < CONTRACT['l1'], (name, t)
        assert abs(ometa.psnr(a, tgt) - ometa.psnr(b, tgt)) < CONTRACT['psnr'], (name, t)
    # SGD-type rules cannot flip; sign-like rules flip where |g| is below conv rounding noise (78 of 3.8M after one
    # VoxelFlow step in round 1): a fraction, not a population
    if opt != 'SGD':
        assert flipped <= 1e-2 * total, (name, flipped, total)     # measured: 0.29 % (VoxelFlow, Meta-SGD + Adamax, 2 steps x 2 tasks)


@pytest.mark.parametrize("name", ['sepconv_lslr_sgd_2step', 'voxelflow_lslr_sgd_2step', 'c1_cain_lslr_sgd'])
def test_fused_support_pair_equals_two_single_passes(name):
    _, l1, p1, _, r1 = run_case(name, 'train', fuse=1)
    _, l0, p0, _, r0 = run_case(name, 'train', fuse=0)
    assert abs(l1['loss'].item() - l0['loss'].item()) <= (2e-4 if 'voxelflow' in name else 2e-5) * abs(l0['loss'].item())
    for a, b in zip(p1, p0):
        assert (a - b).abs().mean().item() < (1e-4 if 'voxelflow' in name else 1e-5)
    for k in r0['outer_grad_fp']:
        assert_fp_close(r1['outer_grad_fp'][k], r0['outer_grad_fp'][k], 3e-2 if 'voxelflow' in name else 1e-3, k)


# ---------------------------------------------------------------------------------------------
# fused multi-tensor rules against the reference classes' outputs (tests/golden/rules.npz)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["lslr", "metasgd"])
@pytest.mark.parametrize("opt", ["SGD", "Adam", "Adamax"])
def test_fused_rules_match_reference(kind, opt):
    g = golden("rules")
    names = ['a.weight', 'a.bias', 'b.weight', 'c.weight']
    w = {k: torch.from_numpy(g['w0/' + k]).to(DEV) for k in names}
    if kind == 'lslr':
        rule = LSLRGradientDescentLearningRule(device=DEV, optimizer=opt, total_num_inner_loop_steps=3,
                                               use_learnable_learning_rates=False, init_learning_rate=0.01)
    else:
        rule = MetaSGDLearningRule(device=DEV, optimizer=opt, init_learning_rate=0.01)
    rule.initialize(w)
    with torch.no_grad():
        for k, p in rule.names_learning_rates_dict.items():
            p.copy_(torch.from_numpy(g['lr/%s/%s' % (kind, k)]))
    rule.initialize_state()
    for t in range(3):
        grads = {k: torch.from_numpy(g['g%d/%s' % (t, k)]).to(DEV) for k in w}
        if t >= 1 and 'c.weight' in grads:
            grads['c.weight'] = None
        with torch.no_grad():
            w = rule.update_params(w, grads, t)
        assert ('c.weight' in w) == (t == 0)
        for k, v in w.items():
            want = g['out/%s/%s/%d/%s' % (kind, opt, t, k)]
            assert np.abs(v.cpu().numpy() - want).max() <= 1e-6 * max(1.0, np.abs(want).max()), (kind, opt, t, k)


@pytest.mark.parametrize("kind", ["lslr", "metasgd"])
@pytest.mark.parametrize("opt", ["SGD", "Adam", "Adamax"])
def test_fused_rule_lr_gradients_match_oracle_autograd(kind, opt):
    """d(sum c*w_T)/d lr and /d w0 through 3 fused steps == autograd through the oracle's formulas.
    (Adam/Adamax + learnable lr + >= 2 steps is a configuration the reference itself cannot train.)"""
    gen = torch.Generator().manual_seed(11)
    shapes = {'p.weight': (6, 5, 3, 3), 'p.bias': (6,), 'q.weight': (4100,)}
    w0 = {k: torch.randn(s, generator=gen) for k, s in shapes.items()}
    grads = [{k: torch.randn(s, generator=gen) for k, s in shapes.items()} for _ in range(3)]
    coef = {k: torch.randn(s, generator=gen) for k, s in shapes.items()}
    # oracle, CPU autograd
    wo = {k: v.clone().requires_grad_() for k, v in w0.items()}
    lrs = orules.init_lrs(kind, wo, 0.01, num_steps=3, learnable=True)
    with torch.no_grad():
        for i, p in enumerate(lrs.values()):
            p.mul_(1.0 + 0.1 * i)
    st, cur = orules.RuleState(), dict(wo)
    for t in range(3):
        cur = orules.update_params(kind, opt, cur, grads[t], lrs, t, st)
    sum((coef[k] * cur[k]).sum() for k in cur).backward()
    # product, fused kernels
    wd = {k: v.clone().to(DEV).requires_grad_() for k, v in w0.items()}
    if kind == 'lslr':
        rule = LSLRGradientDescentLearningRule(device=DEV, optimizer=opt, total_num_inner_loop_steps=3,
                                               use_learnable_learning_rates=True, init_learning_rate=0.01)
    else:
        rule = MetaSGDLearningRule(device=DEV, optimizer=opt, init_learning_rate=0.01)
    rule.initialize(wd)
    with torch.no_grad():
        for i, p in enumerate(rule.names_learning_rates_dict.values()):
            p.mul_(1.0 + 0.1 * i)
    rule.initialize_state()
    cur = dict(wd)
    for t in range(3):
        cur = rule.update_params(cur, {k: v.to(DEV) for k, v in grads[t].items()}, t)
    sum((coef[k].to(DEV) * cur[k]).sum() for k in cur).backward()
    for k in w0:
        assert torch.allclose(wd[k].grad.cpu(), wo[k].grad, rtol=1e-5, atol=1e-6)
        a, b = rule.names_learning_rates_dict[orules.lr_key(k)].grad.cpu(), lrs[orules.lr_key(k)].grad
        assert (a - b).abs().max() <= 2e-5 * b.abs().max() + 1e-6, (kind, opt, k)


def test_l2f_mean_and_scale_kernels():
    gen = torch.Generator().manual_seed(5)
    ts = [torch.randn(s, generator=gen) for s in [(192, 192, 3, 3), (192,), (12, 192, 1, 1), (5,), (4097,)] * 30]
    emb = hip_ops.mt_mean([t.to(DEV) for t in ts])
    want = torch.stack([t.double().mean() for t in ts]).float()
    assert (emb.cpu() - want).abs().max() < 1e-6
    gamma = torch.rand(len(ts), generator=gen).to(DEV).requires_grad_()
    wd = [t.to(DEV).requires_grad_() for t in ts]
    outs = hip_ops.mt_scale(gamma, wd)
    co = [torch.randn(t.shape, generator=gen) for t in ts]
    sum((c.to(DEV) * o).sum() for c, o in zip(co, outs)).backward()
    for i, (o, t) in enumerate(zip(outs, ts)):
        assert torch.allclose(o.detach().cpu(), gamma[i].item() * t, rtol=1e-6, atol=1e-7)
        assert torch.allclose(wd[i].grad.cpu(), gamma[i].item() * co[i], rtol=1e-6, atol=1e-7)
    want_g = torch.stack([(c.double() * t.double()).sum() for c, t in zip(co, ts)]).float()
    assert (gamma.grad.cpu() - want_g).abs().max() <= 1e-4 * want_g.abs().max()


def test_product_path_uses_the_hip_library_and_fails_loudly_without_it(monkeypatch):
    assert _hip.lib().savfi_version() == _hip.ABI_VERSION
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", "/nonexistent/libsavfi_hip.so")
    with pytest.raises(_hip.SavfiHipError):
        hip_ops.l1_loss(torch.zeros(4, device=DEV), torch.zeros(4, device=DEV))


@pytest.mark.parametrize("model", ["sepconv", "cain", "rrin", "superslomo"])
def test_run_test_iter_matches_reference_fixture(model):
    """--mode test: adapt on a 4-frame clip and interpolate between frames 1 and 2 (reference run_test_iter)."""
    g = golden("test_mode")
    over = dict(eval(str(g[model + '_args'])))
    over.setdefault('number_of_training_steps_per_iter', over['number_of_evaluation_steps_per_iter'])
    system = build_system(model, dict(over, mode='test'))
    frames = synthetic.septuplet_batch(2, 64, 64, model=model, frames=4)
    preds = system.run_test_iter(data_batch=frames)
    assert len(preds) == 2 and preds[0].shape == (3, 64, 64)
    assert np.abs(torch.stack(preds).cpu().numpy() - g[model + '_preds']).mean() < 1e-4


def test_second_order_cain_matches_oracle():
    """--second_order (create_graph=True): the update goes through composed device ops, pixel shuffle stays
    differentiable, the loss uses composed ops -- outer gradients must equal the oracle's second-order run."""
    from oracle import meta as ometa
    from tests.helpers import oracle_base
    over = dict(optimizer='SGD', inner_lr=1e-2, loss='1*MSE', number_of_training_steps_per_iter=2,
                number_of_evaluation_steps_per_iter=2, second_order=True, first_order_to_second_order_epoch=-1)
    system = build_system('cain', over)
    rec = observe(system)
    frames = synthetic.septuplet_batch(1, 64, 64, model='cain')
    losses, _, _ = system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=False)
    base = oracle_base('cain')
    names_w = {n: base[n] for n in ometa.inner_param_names([(n, p) for n, p in base.items() if p.is_floating_point()])}
    lrs = orules.init_lrs('lslr', names_w, 1e-2, num_steps=2)
    torch.set_num_threads(16)
    res = ometa.run_iteration('cain', base, frames, rule='lslr', optimizer='SGD', lrs=lrs, num_steps=2, loss='MSE',
                              training=True, second_order=True)
    res['loss'].backward()
    assert abs(losses['loss'].item() - res['loss'].item()) <= 5e-5 * abs(res['loss'].item())
    checked = 0
    for n, p in base.items():
        key = 'net.' + n
        if p.requires_grad and p.grad is not None and key in rec['outer_grad_fp']:
            assert_fp_close(rec['outer_grad_fp'][key], fp(p.grad), 2e-3, ('second-order', n))
            checked += 1
    assert checked == 494


def test_second_order_voxelflow_matches_oracle():
    """--second_order through the VoxelFlow tail: set_double_backward(True) swaps the fused warp (first order only, like ATen's
    grid_sample) for composed device ops; outer gradients must equal the oracle's second-order run through its gather-based
    statement of the tail (oracle/torch_ops.voxel_warp_blend_written_out).  Seeded weights at the model's own scale ('smooth':
    sub-pixel flows -- second derivatives of a chaotic flow field are not a test)."""
    from oracle import meta as ometa, torch_ops as O
    from tests.helpers import oracle_base
    over = dict(optimizer='SGD', inner_lr=1e-2, loss='1*MSE', number_of_training_steps_per_iter=2,
                number_of_evaluation_steps_per_iter=2, second_order=True, first_order_to_second_order_epoch=-1)
    system = build_system('voxelflow', dict(over, weight_recipe='smooth'))
    rec = observe(system)
    frames = synthetic.septuplet_batch(1, 64, 64, model='voxelflow')
    losses, _, _ = system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=False)
    base = oracle_base('voxelflow', recipe='smooth')
    names_w = {n: base[n] for n in ometa.inner_param_names([(n, p) for n, p in base.items() if p.is_floating_point()])}
    lrs = orules.init_lrs('lslr', names_w, 1e-2, num_steps=2)
    torch.set_num_threads(16)
    res = ometa.run_iteration('voxelflow', base, frames, rule='lslr', optimizer='SGD', lrs=lrs, num_steps=2, loss='MSE',
                              training=True, second_order=True, forward_kwargs=dict(warp=O.voxel_warp_blend_written_out))
    res['loss'].backward()
    assert abs(losses['loss'].item() - res['loss'].item()) <= 5e-5 * abs(res['loss'].item())
    checked = 0
    for n, p in base.items():
        key = 'net.' + n
        if p.requires_grad and p.grad is not None and key in rec['outer_grad_fp']:
            assert_fp_close(rec['outer_grad_fp'][key], fp(p.grad), 2e-3, ('second-order', n))
            checked += 1
    assert checked >= 20


# ---------------------------------------------------------------------------------------------
# concurrent tasks (--task_streams 2): one Python thread + HIP stream per task, same results as the sequential loop
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ['sepconv_msl_learnable_2step', 'voxelflow_lslr_sgd_2step', 'superslomo_lslr_sgd_2step'])   # 2-task fixtures
@pytest.mark.parametrize("phase", ["train", "val"])
def test_concurrent_tasks_match_reference_fixture(name, phase):
    g = golden("system_" + name)
    model = str(g['model'])
    assert int(g['B']) == 2
    over = dict(parse_case_args(g), task_streams=2, task_batch=0)
    system = build_system(model, over)
    frames = synthetic.septuplet_batch(int(g['B']), int(g['H']), int(g['W']), model=model)
    rec_outer = {}
    system.optimizer.step = lambda *a, **k: rec_outer.update(
        {n: helpers_fp(p.grad) for n, p in system.named_parameters() if p.requires_grad and p.grad is not None})
    if phase == 'train':
        losses, preds, metrics = system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=True)
    else:
        losses, preds, metrics = system.run_validation_iter(data_batch=frames)
    torch.cuda.synchronize()
    tol = TOL[name]
    want_loss = float(g[phase + '_loss'])
    assert abs(losses['loss'].item() - want_loss) <= tol['loss'] * abs(want_loss)
    got = torch.stack([p.squeeze(0) for p in preds]).cpu().numpy()
    assert np.abs(got - g[phase + '_preds']).mean() < tol['l1']
    assert abs(metrics['psnr'].avg - float(g[phase + '_psnr'])) < tol['psnr']
    if phase == 'train':
        rows = dict(zip(list(g['outer_grad_fp_0_keys']), g['outer_grad_fp_0']))
        assert set(rec_outer) == set(rows)
        for k, row in rows.items():
            assert_fp_close(rec_outer[k], row, tol['outer'], (name, 'outer', k))


@pytest.mark.parametrize("name", ['sepconv_lslr_sgd_2step', 'sepconv_msl_learnable_2step', 'c1_cain_lslr_sgd', 'cain_l2f',
                                  'rrin_lslr_sgd_2step', 'superslomo_lslr_sgd_2step'])
def test_weight_gradients_on_a_side_stream_match_reference_fixture(name):
    """--wgrad_overlap 1: the weight gradients of the support passes run beside the data-gradient chain; same fixtures,
    same gates (per-step gradient / weight fingerprints included)."""
    g = golden("system_" + name)
    model = str(g['model'])
    system = build_system(model, dict(parse_case_args(g), wgrad_overlap=1, task_batch=0))
    rec = observe(system)
    frames = synthetic.septuplet_batch(int(g['B']), int(g['H']), int(g['W']), model=model)
    losses, preds, metrics = system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=True)
    torch.cuda.synchronize()
    tol = TOL[name]
    want_loss = float(g['train_loss'])
    assert abs(losses['loss'].item() - want_loss) <= tol['loss'] * abs(want_loss)
    got = torch.stack([p.squeeze(0) for p in preds]).cpu().numpy()
    assert np.abs(got - g['train_preds']).mean() < tol['l1']
    assert list(g['train_n_live']) == rec['n_live']
    for i, d in enumerate(rec['grad_fp']):
        for k, row in zip(list(g['train_grad_fp_%d_keys' % i]), g['train_grad_fp_%d' % i]):
            assert_fp_close(d[k], row, tol['g'], (name, 'g', i, k))
    for i, d in enumerate(rec['weight_fp']):
        for k, row in zip(list(g['train_weight_fp_%d_keys' % i]), g['train_weight_fp_%d' % i]):
            assert_fp_close(d[k], row, tol['w'], (name, 'w', i, k), extra_abs=knee_allowance(g, 'train', i, k, tol['g']))


@pytest.mark.parametrize("optimizer,metasgd", [("Adam", False), ("Adamax", True)])
def test_concurrent_tasks_keep_rule_state_per_task(optimizer, metasgd):
    """Stateful inner rules (moments, step counts) under --task_streams 2: every task sees its own state.  Checked on the
    step counters (exact), not on pixels: Adam-type steps amplify MIOpen's run-to-run solver differences."""
    import threading
    over = dict(optimizer=optimizer, metasgd=metasgd, inner_lr=1e-4, loss='1*L1', batch_size=4, task_streams=2, task_batch=0,
                number_of_training_steps_per_iter=2, number_of_evaluation_steps_per_iter=2)
    frames = synthetic.septuplet_batch(4, 64, 64, model='cain')
    system = build_system('cain', over)
    rule = system.inner_loop_optimizer
    orig, seen = rule.update_params, []

    def update_params(names_weights_dict, names_grads_wrt_params_dict, num_step, **kw):
        out = orig(names_weights_dict=names_weights_dict, names_grads_wrt_params_dict=names_grads_wrt_params_dict,
                   num_step=num_step, **kw)
        seen.append((threading.get_ident(), num_step, sorted({st['step'] for st in rule.state.values()})))
        return out
    rule.update_params = update_params
    losses, preds, _ = system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=False)
    torch.cuda.synchronize()
    assert len(seen) == 4 * 2 and len({tid for tid, _, _ in seen}) == 2          # 4 tasks x 2 steps on two worker threads
    for tid, num_step, steps in seen:
        assert steps == [num_step + 1], (tid, num_step, steps)                   # a shared state would count both tasks
    assert np.isfinite(losses['loss'].item()) and all(torch.isfinite(p).all() for p in preds)


@pytest.mark.parametrize("name", ['sepconv_msl_learnable_2step', 'voxelflow_lslr_sgd_2step', 'superslomo_lslr_sgd_2step'])
def test_graph_replays_on_two_task_streams_match_reference_fixture(name):
    """--graph_inner_loop 1 --task_streams 2: one graph set per stream, replayed from two threads; outer gradients merged."""
    g = golden("system_" + name)
    model = str(g['model'])
    system = build_system(model, dict(parse_case_args(g), graph_inner_loop=1, task_streams=2, task_batch=0))
    rec_outer = {}
    system.optimizer.step = lambda *a, **k: rec_outer.update(
        {n: helpers_fp(p.grad) for n, p in system.named_parameters() if p.requires_grad and p.grad is not None})
    frames = synthetic.septuplet_batch(int(g['B']), int(g['H']), int(g['W']), model=model)
    losses, preds, metrics = system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=True)
    torch.cuda.synchronize()
    tol = TOL[name]
    want_loss = float(g['train_loss'])
    assert abs(losses['loss'].item() - want_loss) <= tol['loss'] * abs(want_loss)
    got = torch.stack([p.squeeze(0) for p in preds]).cpu().numpy()
    assert np.abs(got - g['train_preds']).mean() < tol['l1']
    assert abs(metrics['psnr'].avg - float(g['train_psnr'])) < tol['psnr']
    rows = dict(zip(list(g['outer_grad_fp_0_keys']), g['outer_grad_fp_0']))
    for k, row in rows.items():
        if abs(row[1]) > 0:
            assert k in rec_outer, k
            assert_fp_close(rec_outer[k], row, tol['outer'], (name, 'outer', k))


# ---------------------------------------------------------------------------------------------
# hipGraph-captured inner loop (--graph_inner_loop 1): same results as the eager loop and the fixtures
# ---------------------------------------------------------------------------------------------
GRAPH_CASES = ['sepconv_lslr_sgd_2step', 'sepconv_msl_learnable_2step', 'sepconv_metasgd_adamax_2step',
               'voxelflow_lslr_sgd_2step', 'c1_cain_lslr_sgd', 'cain_lslr_adam_1step', 'rrin_lslr_sgd_2step', 'superslomo_lslr_sgd_2step',
               'cain_l2f']


@pytest.mark.parametrize("name", GRAPH_CASES)
@pytest.mark.parametrize("phase", ["train", "val"])
def test_graphed_inner_loop_matches_reference_fixture(name, phase, monkeypatch):
    from meta_interpolation_amd import graph_inner_loop
    monkeypatch.setattr(graph_inner_loop, 'GRAPH_L2F', True)       # the opt-in graphed L2F (cain_l2f) is held to its fixture too
    tol = TOL[name]
    g = golden("system_" + name)
    model = str(g['model'])
    system = build_system(model, dict(parse_case_args(g), graph_inner_loop=1, task_batch=0, task_streams=1))
    rec = {}
    system.optimizer.step = lambda *a, **k: rec.update(
        {n: fp(p.grad) for n, p in system.named_parameters() if p.requires_grad and p.grad is not None})
    frames = synthetic.septuplet_batch(int(g['B']), int(g['H']), int(g['W']), model=model)
    for rep in range(2):      # second call replays the already captured graphs
        if phase == 'train':
            losses, preds, metrics = system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=True)
        else:
            losses, preds, metrics = system.run_validation_iter(data_batch=frames)
        torch.cuda.synchronize()
        assert len(system._graphs) == 1
        want_loss = float(g[phase + '_loss'])
        assert abs(losses['loss'].item() - want_loss) <= tol['loss'] * abs(want_loss)
        got = torch.stack([p.squeeze(0) for p in preds]).cpu().numpy()
        assert np.abs(got - g[phase + '_preds']).mean() < tol['l1']
        assert abs(metrics['psnr'].avg - float(g[phase + '_psnr'])) < tol['psnr']
        if phase == 'train':
            rows = dict(zip(list(g['outer_grad_fp_0_keys']), g['outer_grad_fp_0']))
            # tensors the plugin never reads from the fast dict have no graphed lr gradient (it is exactly zero
            # in the reference as well: their updated copies are never used)
            for k, row in rows.items():
                if k in rec:
                    assert_fp_close(rec[k], row, tol['outer'], (name, 'outer', k))
                else:
                    assert abs(row[1]) == 0.0, (name, 'missing outer grad', k)


def test_sepconv_trains_from_a_vimeo_directory_through_the_frame_stager(tmp_path, monkeypatch):
    """main.py end to end on the GPU: VimeoSeptuplet reader -> FrameStager (uint8 H2D + savfi_frames_u8_to_f32 on a
    side stream) -> SceneAdaptiveInterpolation -> ExperimentBuilder (train iterations, validation sweep, checkpoint)."""
    import os
    from meta_interpolation_amd.config import default_args
    from meta_interpolation_amd.data import MetaLearningSystemDataLoader
    from meta_interpolation_amd.experiment_builder import ExperimentBuilder
    from meta_interpolation_amd.meta_learning_system import SceneAdaptiveInterpolation
    monkeypatch.chdir(tmp_path)
    root = synthetic.write_fake_vimeo(str(tmp_path / 'vimeo'))
    args = default_args(model='sepconv', num_gpu=1, batch_size=2, number_of_training_steps_per_iter=1,
                        number_of_evaluation_steps_per_iter=1, optimizer='SGD', loss='1*L1', inner_lr=1e-5, dataset='vimeo90k',
                        data_root=root, total_iter_per_epoch=2, max_epoch=1, exp_name='vimeo_e2e', num_workers=3)
    from meta_interpolation_amd.meta_learning_system import MODEL_REGISTRY
    net = MODEL_REGISTRY['sepconv'](args, False)          # no pretrained_models/*.pth here: seeded weights
    synthetic.load_seeded_weights(net, 'sepconv')
    system = SceneAdaptiveInterpolation(args, net=net.cuda())
    eb = ExperimentBuilder(args, MetaLearningSystemDataLoader, system)
    assert eb.data.stager is not None
    eb.run_experiment()
    torch.cuda.synchronize()
    assert eb.state['current_iter'] == 2 and eb.epoch == 1
    assert os.path.exists(os.path.join('checkpoint', 'vimeo_e2e', 'checkpoint.pth'))
    assert all(torch.isfinite(p).all() for p in system.parameters())


# ---------------------------------------------------------------------------------------------
# tasks in lockstep (--task_batch T): one launch per layer for all tasks of a meta-batch, per-task fast weights
# ---------------------------------------------------------------------------------------------
@pytest.fixture
def lockstep_for(monkeypatch):
    """Lockstep switch for the lockstep tests.  Every shipped plugin opts in (VoxelFlow since round 3: its layers run on the
    direct split-bf16 kernels with per-task filter sets).  For VoxelFlow the Winograd task kernels are additionally fenced off:
    F(2x2,3x3) rounds ~3x coarser than a direct convolution and its flow-to-pixel map turns that into 1e-3 of a step-0 gradient
    fingerprint -- the product never sends its layers there (MetaConv2dLayer(direct=True)), the fence keeps it that way here."""
    def apply(system, model):
        system.net.lockstep_tasks = True
        if model == 'voxelflow':
            monkeypatch.setattr(hip_ops, 'TASKS_MIN_TILES_FWD', 10 ** 9)
            monkeypatch.setattr(hip_ops, 'TASKS_MIN_TILES_BWD', 10 ** 9)
            monkeypatch.setattr(hip_ops, 'TASKS_WGRAD_MIN_PIXELS', 10 ** 9)
    return apply


@pytest.mark.parametrize("name", ['sepconv_msl_learnable_2step', 'voxelflow_lslr_sgd_2step', 'voxelflow_metasgd_adamax_2step',
                                  'superslomo_lslr_sgd_2step',
          # 128 x 128 twins (deepest maps 4 x 4): the OUTER-gradient fingerprints at the plain 1e-3 gate
          'sepconv_msl_learnable_2step_128', 'superslomo_lslr_sgd_2step_128'])          # the 2-task fixtures
@pytest.mark.parametrize("phase", ["train", "val"])
def test_lockstep_tasks_match_reference_fixture(name, phase, lockstep_for):
    g = golden("system_" + name)
    model = str(g['model'])
    assert int(g['B']) == 2
    system = build_system(model, dict(parse_case_args(g), task_batch=2))
    lockstep_for(system, model)
    calls = []
    orig = system._lockstep_body
    system._lockstep_body = lambda *a, **k: (calls.append(len(a[1])), orig(*a, **k))[1]
    rec_outer = {}
    system.optimizer.step = lambda *a, **k: rec_outer.update(
        {n: helpers_fp(p.grad) for n, p in system.named_parameters() if p.requires_grad and p.grad is not None})
    frames = synthetic.septuplet_batch(2, int(g['H']), int(g['W']), model=model)
    if phase == 'train':
        losses, preds, metrics = system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=True)
    else:
        losses, preds, metrics = system.run_validation_iter(data_batch=frames)
    torch.cuda.synchronize()
    assert calls == [2]
    tol = tolerances(name, phase)
    want_loss = float(g[phase + '_loss'])
    assert abs(losses['loss'].item() - want_loss) <= tol['loss'] * abs(want_loss)
    got = torch.stack([p.squeeze(0) for p in preds]).cpu().numpy()
    assert np.abs(got - g[phase + '_preds']).mean() < tol['l1']
    assert abs(metrics['psnr'].avg - float(g[phase + '_psnr'])) < tol['psnr']
    assert abs(float(metrics['ssim'].avg) - float(g[phase + '_ssim'])) < tol['ssim']
    for k in ('L1', 'MSE'):
        if phase + '_part_' + k in g.files:
            assert abs(float(losses[k]) - float(g[phase + '_part_' + k])) <= 5 * tol['loss'] * abs(float(g[phase + '_part_' + k]))
    if phase == 'train':
        rows = dict(zip(list(g['outer_grad_fp_0_keys']), g['outer_grad_fp_0']))
        for k, row in rows.items():
            if abs(row[1]) > 0:       # tensors the plugin never routes have exactly-zero lr gradients in the reference
                assert k in rec_outer, k
                assert_fp_close(rec_outer[k], row, tol['outer'], (name, 'outer', k))


@pytest.mark.parametrize("model,over", [
    ('sepconv', dict(optimizer='SGD', inner_lr=1e-3, loss='1*L1')),
    ('cain', dict(optimizer='Adam', inner_lr=1e-4, loss='1*L1', metasgd=True)),
    ('cain', dict(optimizer='SGD', inner_lr=1e-3, loss='1*L1', attenuate=True)),
    ('voxelflow', dict(optimizer='SGD', inner_lr=1e-3, loss='1*MSE')),
    ('rrin', dict(optimizer='SGD', inner_lr=1e-3, loss='1*L1'))])
def test_lockstep_equals_the_sequential_loop(model, over, lockstep_for):
    """5 tasks in groups of up to 4 (so one lockstep group of 4 and a sequential straggler) against the plain task loop:
    losses, predictions, PSNR and every outer gradient."""
    over = dict(over, number_of_training_steps_per_iter=2, number_of_evaluation_steps_per_iter=2, batch_size=5)
    frames = synthetic.septuplet_batch(5, 64, 64, model=model)
    got = {}
    for tb in (0, 4):
        system = build_system(model, dict(over, task_batch=tb))
        lockstep_for(system, model)
        grads = {}
        system.optimizer.step = lambda *a, **k: grads.update(
            {n: p.grad.detach().clone() for n, p in system.named_parameters() if p.requires_grad and p.grad is not None})
        losses, preds, metrics = system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=True)
        torch.cuda.synchronize()
        got[tb] = (losses['loss'].item(), torch.stack([p.squeeze(0) for p in preds]), metrics['psnr'].avg, grads)
    (l0, p0, s0, g0), (l1, p1, s1, g1) = got[0], got[4]
    sign_like = over['optimizer'] != 'SGD'
    # VoxelFlow: its 5x5 layers run on another MIOpen solver when grouped and the flow-to-pixel map amplifies conv rounding (the
    # reference moves by 7e-5..9e-4 in loss against itself: tests/golden/sensitivity.npz); measured here: 4.1e-4
    assert abs(l0 - l1) <= (1e-3 if model == 'voxelflow' else 2e-5) * abs(l0)
    # VoxelFlow: a self-comparison of two summation orders (the weight-gradient partial blocks are cut differently for 8 and 2
    # samples) under its rounding amplification: the reference moves by 6.9e-5 pixel L1 / 3.3e-4 dB against ITSELF on the 2-task
    # fixture (tests/golden/sensitivity.npz); 3x that here
    l1_lim, psnr_lim = ((3 * 6.9e-5, 3 * 3.4e-4) if model == 'voxelflow' else (1e-4, 1e-3))
    assert (p0 - p1).abs().mean().item() < max(l1_lim, 1e-4) and abs(s0 - s1) < max(psnr_lim, 1e-3)
    for k, v in g0.items():
        if v.abs().sum().item() == 0:
            continue
        assert k in g1, k
        lim = 5e-2 if (sign_like or model == 'voxelflow') else 2e-3
        if over.get('attenuate'):
            # L2F: the attenuation is a function of the embedding pass's gradients, and the reference's own outer gradients move
            # by 2.65e-3 under another conv summation order on the cain_l2f fixture (tests/golden/sensitivity.npz, `outer`; its 16x16
            # maps run on MIOpen here, whose solver for a batch of 4 is not the one for a single sample).  Seen: 3.5e-3 on headConv
            # in one process of five, < 2e-3 in the others; gate at 3 x the reference's self-spread.
            lim = max(lim, 3 * float(_SENS['cain_l2f/train'][:, _SENS_COL['outer']].max()))
        # absolute floor = helpers.FP_ATOL: CAIN's channel-attention squeeze layers (conv_du.0: 12 x 192 weights, 12 biases) have gradient
        # abs-sums of 2e-7 .. 5e-7 where the body convolutions have 1e+1 -- sums of O(1) activations at fp32 leave 1e-9 of noise there
        assert (g1[k] - v).abs().sum().item() <= lim * v.abs().sum().item() + FP_ATOL, (k, (g1[k] - v).abs().sum().item(), v.abs().sum().item())


@pytest.mark.parametrize("name", ['sepconv_msl_learnable_2step', 'voxelflow_lslr_sgd_2step', 'voxelflow_metasgd_adamax_2step',
                                  'superslomo_lslr_sgd_2step',
          # 128 x 128 twins (deepest maps 4 x 4): the OUTER-gradient fingerprints at the plain 1e-3 gate
          'sepconv_msl_learnable_2step_128', 'superslomo_lslr_sgd_2step_128'])
@pytest.mark.parametrize("phase", ["train", "val"])
def test_graphed_lockstep_tasks_match_reference_fixture(name, phase, lockstep_for):
    """--graph_inner_loop 1 --task_batch 2: the lockstep pass replayed from hipGraphs (stacked static buffers)."""
    g = golden("system_" + name)
    model = str(g['model'])
    system = build_system(model, dict(parse_case_args(g), graph_inner_loop=1, task_batch=2))
    lockstep_for(system, model)
    rec_outer = {}
    system.optimizer.step = lambda *a, **k: rec_outer.update(
        {n: helpers_fp(p.grad) for n, p in system.named_parameters() if p.requires_grad and p.grad is not None})
    frames = synthetic.septuplet_batch(2, int(g['H']), int(g['W']), model=model)
    tol = tolerances(name, phase)
    for rep in range(2):          # the second call replays the captured graphs
        if phase == 'train':
            losses, preds, metrics = system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=True)
        else:
            losses, preds, metrics = system.run_validation_iter(data_batch=frames)
        torch.cuda.synchronize()
        assert len(system._graphs) == 1 and next(iter(system._graphs.values())).T == 2
        want_loss = float(g[phase + '_loss'])
        assert abs(losses['loss'].item() - want_loss) <= tol['loss'] * abs(want_loss)
        got = torch.stack([p.squeeze(0) for p in preds]).cpu().numpy()
        assert np.abs(got - g[phase + '_preds']).mean() < tol['l1']
        assert abs(metrics['psnr'].avg - float(g[phase + '_psnr'])) < tol['psnr']
        if phase == 'train':
            rows = dict(zip(list(g['outer_grad_fp_0_keys']), g['outer_grad_fp_0']))
            for k, row in rows.items():
                if abs(row[1]) > 0:
                    assert k in rec_outer, k
                    assert_fp_close(rec_outer[k], row, tol['outer'], (name, 'outer', k))


def test_default_execution_mode_policy():
    """config.py defaults (--graph_inner_loop -1, --task_batch 8): a rank with ONE task replays hipGraphs (launch-bound pass), a rank
    with several adapts them in lockstep in the eager loop, L2F stays eager (its graphed form is opt-in: graph_inner_loop.GRAPH_L2F) -- all with the fixture's numbers."""
    for name, want_graphs, want_lockstep in (('c1_cain_lslr_sgd', 1, 0), ('sepconv_msl_learnable_2step', 0, 1), ('cain_l2f', 0, 0),
                                             ('voxelflow_lslr_sgd_2step', 0, 1)):
        g = golden("system_" + name)
        model = str(g['model'])
        system = build_system(model, dict(parse_case_args(g), graph_inner_loop=-1, task_streams=-1))
        calls = []
        orig = system._lockstep_body
        system._lockstep_body = lambda *a, **k: (calls.append(len(a[1])), orig(*a, **k))[1]
        frames = synthetic.septuplet_batch(int(g['B']), int(g['H']), int(g['W']), model=model)
        losses, preds, metrics = system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=True)
        torch.cuda.synchronize()
        assert len(system._graphs) == want_graphs and len(calls) == want_lockstep, (name, len(system._graphs), calls)
        tol = tolerances(name, 'train')
        assert abs(losses['loss'].item() - float(g['train_loss'])) <= tol['loss'] * abs(float(g['train_loss']))
        got = torch.stack([p.squeeze(0) for p in preds]).cpu().numpy()
        assert np.abs(got - g['train_preds']).mean() < tol['l1']
        assert abs(metrics['psnr'].avg - float(g['train_psnr'])) < tol['psnr']
