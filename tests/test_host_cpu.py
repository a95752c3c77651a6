"""CPU (-m "not gpu"): host logic that needs no kernel launch -- the C-ABI library loads and exports
every declared symbol, flags / parameter names / state_dict keys match the reference's contract, the
fast-weight routing, the MSL vector, metrics, the synthetic recipe, loud failure on CPU tensors."""
import os
import ctypes
import types

import numpy as np
import pytest
import torch

from meta_interpolation_amd import _hip, hip_ops, model_utils, synthetic
from meta_interpolation_amd.config import default_args, get_args
from meta_interpolation_amd.meta_learning_system import MODEL_REGISTRY, SceneAdaptiveInterpolation
from tests.helpers import build_plugin, golden


def test_library_loads_and_exports_every_declared_symbol():
    declared = _hip.declared_symbols()
    assert len(declared) >= 14 and "savfi_sepconv_fwd_f32" in declared
    handle = ctypes.CDLL(_hip.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), "libsavfi_hip.so does not export " + name
    assert set(declared) == set(_hip._PROTOTYPES), "ctypes prototypes out of sync with include/savfi_hip.h"
    assert _hip.lib().savfi_version() == _hip.ABI_VERSION


def test_ops_refuse_cpu_tensors_loudly():
    from meta_interpolation_amd.sepconv.sepconv_op.sepconv import FunctionSepconv
    with pytest.raises(NotImplementedError):
        FunctionSepconv.apply(torch.zeros(1, 3, 52, 52), torch.zeros(1, 51, 2, 2), torch.zeros(1, 51, 2, 2))
    with pytest.raises(NotImplementedError):
        hip_ops.pixel_shuffle(torch.zeros(1, 3, 8, 8), 1 / 8)
    with pytest.raises(NotImplementedError):
        hip_ops.voxel_warp_blend(torch.zeros(1, 6, 8, 8), torch.zeros(1, 3, 8, 8))
    with pytest.raises(NotImplementedError):
        hip_ops.mt_update(_hip.RULE_SGD, _hip.LR_SCALAR, [torch.zeros(3)], [torch.zeros(3)], [torch.tensor(0.1)])
    with pytest.raises(NotImplementedError):
        hip_ops.l1_loss(torch.zeros(3), torch.zeros(3))


def test_flags_keep_reference_names_and_defaults():
    a = default_args()
    expect = dict(model='CAIN', mode='train', loss='1*L1', optimizer='Adam', inner_lr=1e-5, outer_lr=1e-5,
                  batch_size=8, number_of_training_steps_per_iter=1, number_of_evaluation_steps_per_iter=1,
                  learnable_per_layer_per_step_inner_loop_learning_rate=False,
                  enable_inner_loop_optimizable_bn_params=False, second_order=False,
                  first_order_to_second_order_epoch=-1, use_multi_step_loss_optimization=False,
                  multi_step_loss_num_epochs=1, attenuate=False, metasgd=False, num_gpu=1, random_seed=12345,
                  resume=False, pretrained_model=None, weight_decay=1e-4, total_iter_per_epoch=10, eval_iter=10)
    for k, v in expect.items():
        assert getattr(a, k) == v, k
    assert a.cuda is True
    b, rest = get_args(['--model', 'sepconv', '--num_gpu', '0', '--metasgd', '--optimizer', 'Adamax', '--bogus', '1'])
    assert b.model == 'sepconv' and b.cuda is False and b.metasgd and rest == ['--bogus', '1']


@pytest.mark.parametrize("model,count,numel", [("sepconv", 94, 21675452), ("voxelflow", 23, 3821891),
                                               ("cain", 494, 42780432), ("rrin", 162, 19194445),
                                               ("superslomo", 92, 39610473)])
def test_inner_loop_dict_sizes(model, count, numel):
    """94 / 23 / 494 tensors."""
    net = build_plugin(model)
    args = default_args(model=model, num_gpu=0)
    stub = types.SimpleNamespace(args=args)
    d = SceneAdaptiveInterpolation.get_inner_loop_parameter_dict(stub, net.named_parameters())
    assert len(d) == count and sum(p.numel() for p in d.values()) == numel


@pytest.mark.parametrize("case,model", [("system_sepconv_lslr_sgd_2step", "sepconv"),
                                        ("system_voxelflow_metasgd_adamax_2step", "voxelflow"),
                                        ("system_c1_cain_lslr_sgd", "cain"),
                                        ("system_rrin_lslr_sgd_2step", "rrin"),
                                        ("system_superslomo_lslr_sgd_2step", "superslomo")])
def test_parameter_names_match_the_reference(case, model):
    """Names key the lr tables, checkpoints and the fast-weight routing: they must be the reference's."""
    g = golden(case)
    ref_net = {k[4:] for k in g['outer_grad_fp_0_keys'] if str(k).startswith('net.')}
    ours = {n for n, p in build_plugin(model).named_parameters() if p.requires_grad}
    assert ours == ref_net
    ref_step0 = list(g['train_weight_fp_0_keys'])
    assert sorted(ours & set(ref_step0)) == sorted(ref_step0)


def test_state_dict_key_layout():
    args = default_args(model='voxelflow', num_gpu=0, metasgd=True, attenuate=True, optimizer='Adam')
    system = SceneAdaptiveInterpolation(args, net=build_plugin('voxelflow'))
    keys = set(system.state_dict())
    assert 'net.conv1.weight' in keys and 'net.conv1_bn.running_mean' in keys
    assert 'inner_loop_optimizer.names_learning_rates_dict.conv1-weight' in keys
    assert {'attenuator.0.weight', 'attenuator.0.bias', 'attenuator.2.weight', 'attenuator.2.bias', 'gamma_mult'} <= keys
    assert system.mean.shape == (3, 1, 1) and float(system.std[0]) == 127.5
    assert system.optimizer.param_groups[0]['lr'] == args.outer_lr and len(system.optimizer.param_groups) == 3
    system.scheduler.step(1.0)


def test_unknown_model_and_optimizer_raise_like_the_reference():
    with pytest.raises(NotImplementedError):
        SceneAdaptiveInterpolation(default_args(model='nope', num_gpu=0))
    from meta_interpolation_amd.inner_loop_optimizers import LSLRGradientDescentLearningRule
    rule = LSLRGradientDescentLearningRule('cpu', 'RMSprop', 1, False, 0.1)
    with pytest.raises(NotImplementedError):
        rule.update_params({'a': torch.zeros(1)}, {'a': torch.zeros(1)}, 0)


@pytest.mark.parametrize("S,epoch,E", [(5, 0, 10), (5, 3, 10), (5, 50, 10), (1, 0, 1), (3, 2, 4)])
def test_msl_importance_vector_matches_reference(S, epoch, E):
    stub = types.SimpleNamespace(args=types.SimpleNamespace(number_of_training_steps_per_iter=S,
                                                            multi_step_loss_num_epochs=E),
                                 current_epoch=epoch, device=torch.device('cpu'))
    v = SceneAdaptiveInterpolation.get_per_step_loss_importance_vector(stub)
    assert np.array_equal(v.numpy(), golden("ops")['msl_%d_%d_%d' % (S, epoch, E)])


def test_param_view_routing():
    flat = {'a.0.weight': 1, 'a.0.bias': 2, 'a.2.weight': 3, 'b.weight': 4}
    pv = model_utils.as_view(flat)
    assert pv.sub('a').sub(0).leaf('weight') == 1 and pv.sub('a')['2']['weight'] == 3 and pv['b.weight'] == 4
    assert 'a' in pv and 'zzz' not in pv
    with pytest.raises(KeyError):
        pv.sub('a').sub(1).leaf('weight')
    nested = {'a': {'0': {'weight': 1, 'bias': 2}}, 'b': {'weight': 4}}
    assert model_utils.as_view(nested).sub('a').sub('0').leaf('bias') == 2
    assert model_utils.extract_top_level_dict(flat) == {'a': {'0.weight': 1, '0.bias': 2, '2.weight': 3},
                                                        'b': {'weight': 4}}


def test_meta_conv_uses_external_weights_on_cpu():
    conv = model_utils.MetaConv2dLayer(2, 3, 3, 1, 1)
    x = torch.randn(1, 2, 5, 5)
    w, b = torch.randn(3, 2, 3, 3), torch.randn(3)
    got = conv(x, params={'weight': w, 'bias': b})
    assert torch.allclose(got, torch.nn.functional.conv2d(x, w, b, 1, 1))
    seq = model_utils.MetaSequential(conv, torch.nn.ReLU(), model_utils.MetaConv2dLayer(3, 1, 1, 1, 0))
    w2, b2 = torch.randn(1, 3, 1, 1), torch.randn(1)
    y = seq(x, params={'0.weight': w, '0.bias': b, '2.weight': w2, '2.bias': b2})
    assert torch.allclose(y, torch.nn.functional.conv2d(torch.relu(got), w2, b2))


def test_sepconv_padding_rule():
    from meta_interpolation_amd.sepconv.model import MetaNetwork
    assert MetaNetwork.padded_size(256, 448) == (384, 512)
    assert MetaNetwork.padded_size(64, 64) == (128, 128)
    assert MetaNetwork.padded_size(78, 78) == (128, 128) and MetaNetwork.padded_size(79, 79) == (256, 256)


def test_synthetic_recipe_is_deterministic_and_well_formed():
    a = synthetic.septuplet_batch(2, 32, 48, model='sepconv')
    b = synthetic.septuplet_batch(2, 32, 48, model='sepconv')
    assert len(a) == 7 and a[0].shape == (2, 3, 32, 48)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert float(a[0].min()) >= 0 and float(a[0].max()) <= 1
    assert torch.equal(torch.round(a[3] * 255) / 255, a[3])          # PNG-like quantisation
    assert not torch.equal(a[0][0], a[0][1])                          # tasks differ
    assert torch.equal(a[1][0, :, :-1, :-2], a[0][0, :, 1:, 2:])      # frame k+1 = frame k shifted by (1,2)
    v = synthetic.septuplet_batch(1, 16, 16, model='voxelflow')[0]
    assert float(v.min()) >= -1 and float(v.max()) <= 1
    n1, n2 = build_plugin('voxelflow'), build_plugin('voxelflow')
    assert all(torch.equal(p, q) for p, q in zip(n1.state_dict().values(), n2.state_dict().values()))


def test_registry_has_the_plugins():
    assert {'sepconv', 'voxelflow', 'cain', 'rrin', 'superslomo'} <= set(MODEL_REGISTRY)


def test_per_task_state_and_flags_are_thread_local():
    """Concurrent task adaptation (--task_streams): the inner rule's per-task moments / step counts and the
    OWN_PARAMS_CONST flag live per thread."""
    import threading
    from meta_interpolation_amd import model_utils
    from meta_interpolation_amd.inner_loop_optimizers import LSLRGradientDescentLearningRule
    rule = LSLRGradientDescentLearningRule(device=torch.device('cpu'), optimizer='Adam', init_learning_rate=1e-3,
                                           total_num_inner_loop_steps=2, use_learnable_learning_rates=False)
    rule.initialize_state()
    rule.state['k'] = {'step': 7}
    model_utils.set_own_params_const(True)
    seen = {}

    def other():
        seen['state'] = dict(rule.state)                 # a fresh thread starts with an empty state ...
        seen['flag'] = model_utils.own_params_const()    # ... and the flag off
        rule.state['k'] = {'step': 1}
        model_utils.set_own_params_const(True)
    th = threading.Thread(target=other)
    th.start()
    th.join()
    try:
        assert seen == {'state': {}, 'flag': False}
        assert rule.state == {'k': {'step': 7}} and model_utils.own_params_const()
    finally:
        model_utils.set_own_params_const(False)


def test_weight_gradient_overlap_is_off_without_a_gpu_and_counts_weight_uses():
    from meta_interpolation_amd import hip_ops
    hip_ops.set_weight_gradient_overlap(False)
    assert hip_ops.weight_gradient_stream() is None
    hip_ops.join_weight_gradients()                      # no side stream was ever created: a no-op, also on CPU
    w = torch.zeros(1)
    assert hip_ops._weight_use_counter(w)[0] == 1 and hip_ops._weight_use_counter(w)[0] == 2
    hip_ops.set_weight_gradient_overlap(False)           # a new pass starts counting again
    assert hip_ops._weight_use_counter(w)[0] == 1


def test_outer_gradient_accumulators_merge_and_install():
    """graph_inner_loop.OuterGradAccumulator: per-stream accumulators of the graphed two-stream mode sum into one and are
    installed as the mean over the global meta-batch."""
    import types
    from meta_interpolation_amd.graph_inner_loop import OuterGradAccumulator
    theta = {'a': torch.nn.Parameter(torch.zeros(3)), 'b': torch.nn.Parameter(torch.zeros(2))}
    rates = torch.nn.ParameterDict({'a': torch.nn.Parameter(torch.zeros(3)), 'b': torch.nn.Parameter(torch.zeros(3))})
    sysm = types.SimpleNamespace(inner_loop_optimizer=types.SimpleNamespace(names_learning_rates_dict=rates))
    one, two = OuterGradAccumulator(sysm, theta), OuterGradAccumulator(sysm, theta)
    one.add_params(['a', 'b'], [torch.ones(3), torch.full((2,), 2.0)])
    one.add_params(['a'], [torch.ones(3)])                                   # second task on the same stream
    two.add_params(['a'], [torch.full((3,), 4.0)])                            # a task on the other stream ('b' unused there)
    one.lr_rows, one.lr_keys = torch.tensor([[1.0, 2.0], [3.0, 4.0], [0.0, 0.0]]), ['a', 'b']
    two.lr_rows, two.lr_keys = torch.tensor([[1.0, 1.0], [1.0, 1.0], [0.0, 0.0]]), ['a', 'b']
    one.merge(two)
    one.install(num_tasks=4)
    assert torch.equal(theta['a'].grad, torch.full((3,), 6.0 / 4)) and torch.equal(theta['b'].grad, torch.full((2,), 2.0 / 4))
    assert torch.equal(rates['a'].grad, torch.tensor([2.0, 4.0, 0.0]) / 4)    # column of tensor 'a': one value per inner step
    assert torch.equal(rates['b'].grad, torch.tensor([3.0, 5.0, 0.0]) / 4)
    empty = OuterGradAccumulator(sysm, theta)
    empty.merge(two)                                                         # merging into an accumulator that saw no task
    assert torch.equal(empty.param['a'], two.param['a']) and empty.param['a'] is not two.param['a']


# ---------------------------------------------------------------------------------------------
# tasks in lockstep (--task_batch): host logic on CPU with the toy plugin -- stacked fast weights, sample-major batches,
# per-sample criterion, outer gradients through the expand of theta; must equal the sequential task loop
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("msl", [False, True])
@pytest.mark.parametrize("width", [2, 3, 5])
def test_lockstep_tasks_equal_the_sequential_loop(msl, width):
    from tests.helpers import build_toy_system
    B = 5
    frames = synthetic.septuplet_batch(B, 16, 24)
    got = {}
    for tb in (0, width):
        system = build_toy_system(batch=B, msl=msl, task_batch=tb)
        grads = {}
        system.optimizer.step = lambda *a, **k: grads.update(
            {n: p.grad.detach().clone() for n, p in system.named_parameters() if p.grad is not None})
        losses, preds, metrics = system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=True)
        vl, vp, vm = system.run_validation_iter(data_batch=frames)
        got[tb] = (losses, preds, metrics, grads, vl, vp, vm)
    (l0, p0, m0, g0, vl0, vp0, vm0), (l1, p1, m1, g1, vl1, vp1, vm1) = got[0], got[width]
    assert abs(float(l0['loss']) - float(l1['loss'])) < 1e-6 and abs(float(vl0['loss']) - float(vl1['loss'])) < 1e-6
    assert abs(m0['psnr'].avg - m1['psnr'].avg) < 1e-4 and abs(vm0['psnr'].avg - vm1['psnr'].avg) < 1e-4
    assert abs(float(l0['L1']) - float(l1['L1'])) < 1e-6
    for a, b in zip(p0 + vp0, p1 + vp1):
        assert a.shape == b.shape and torch.allclose(a, b, atol=1e-6)
    assert set(g0) == set(g1) and any(k.startswith('inner_loop_optimizer') for k in g0)
    for k in g0:
        assert torch.allclose(g0[k], g1[k], rtol=1e-4, atol=1e-7), k


# ---------------------------------------------------------------------------------------------
# pretrained_models/*_base.pth: the layouts the reference's plugins load when --resume is absent
# (cain/model.py:60-67 `state_dict` with DataParallel's `module.` prefix; sepconv/model.py:247-249 a bare state dict;
#  voxel_flow.py:276-281 `state_dict`; superslomo `state_dictFC` / `state_dictAT`) -- written here, read by the plugins
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model", ["cain", "sepconv", "voxelflow", "rrin", "superslomo"])
def test_base_checkpoint_round_trip(model, tmp_path, monkeypatch):
    import argparse
    from meta_interpolation_amd.config import default_args
    from meta_interpolation_amd.meta_learning_system import MODEL_REGISTRY
    src = build_plugin(model)
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    monkeypatch.chdir(tmp_path)
    os.makedirs('pretrained_models')
    extra = {'args': argparse.Namespace(lr=1e-4), 'epoch': 3}          # the authors' files pickle more than tensors
    if model == 'cain':
        torch.save(dict(extra, state_dict={'module.' + k: v for k, v in sd.items()}), 'pretrained_models/cain_base.pth')
    elif model == 'sepconv':
        torch.save(sd, 'pretrained_models/sepconv_base_l1.pth')
    elif model == 'voxelflow':
        torch.save(dict(extra, state_dict=sd), 'pretrained_models/voxelflow_ft.pth')
    elif model == 'rrin':
        torch.save(sd, 'pretrained_models/rrin_base.pth')
    else:
        torch.save(dict(extra, state_dictFC={k[len('flowComp.'):]: v for k, v in sd.items() if k.startswith('flowComp.')},
                        state_dictAT={k[len('arbTimeFlowIntrp.'):]: v for k, v in sd.items() if k.startswith('arbTimeFlowIntrp.')}),
                   'pretrained_models/superslomo_base.pth')
    args = default_args(model=model, num_gpu=0)
    net = MODEL_REGISTRY[model](args, True)                             # resume=True: "load the base weights"
    got = net.state_dict()
    assert set(got) == set(sd)
    for k, v in sd.items():
        assert torch.equal(got[k], v), k


def test_composed_voxel_warp_for_second_order_matches_the_oracle_and_differentiates_twice():
    """hip_ops.voxel_warp_blend under set_double_backward(True) (what --second_order takes): composed ATen ops == the oracle's tail
    (values, first-order gradients), twice differentiable; the fused op stays the first-order path."""
    import torch
    from meta_interpolation_amd import hip_ops
    from oracle import torch_ops as O
    gen = torch.Generator().manual_seed(11)
    fr = torch.rand(2, 6, 16, 24, generator=gen).requires_grad_()
    x3 = ((torch.rand(2, 3, 16, 24, generator=gen) * 2 - 1) * 0.95).requires_grad_()
    hip_ops.set_double_backward(True)
    try:
        a = hip_ops.voxel_warp_blend(fr, x3)          # CPU tensors: only the composed path accepts them
    finally:
        hip_ops.set_double_backward(False)
    b = O.voxel_warp_blend(fr, x3)
    assert (a - b).abs().max().item() < 2e-6
    ga, gb = torch.autograd.grad(a.pow(2).sum(), [fr, x3], retain_graph=True), torch.autograd.grad(b.pow(2).sum(), [fr, x3])
    for u, v in zip(ga, gb):
        assert (u - v).abs().max().item() < 2e-5 * max(1.0, v.abs().max().item())
    f = torch.rand(1, 6, 5, 6, dtype=torch.double, generator=gen).requires_grad_()
    x = ((torch.rand(1, 3, 5, 6, dtype=torch.double, generator=gen) * 2 - 1) * 0.9).requires_grad_()
    assert torch.autograd.gradgradcheck(hip_ops._voxel_warp_composed, (f, x), eps=1e-7, atol=1e-5)


def test_3x3_weight_gradient_routing_rules(monkeypatch):
    """hip_ops.convk_wgrad_preferred: which weight gradients go to the split-bf16 kernels (pure host logic; the measured table behind the
    rules: profiles/r05_wgrad3_forms.txt) -- and the library agrees on which of them hand out the bias sums."""
    from meta_interpolation_amd import hip_ops, _hip
    pref = hip_ops.convk_wgrad_preferred
    assert pref(5, 6, 64, 256, 256) and pref(7, 32, 32, 256, 448)                       # 5x5 / 7x7: always
    assert pref(3, 128, 128, 96, 128) and pref(3, 64, 51, 137, 236) and pref(3, 51, 51, 256, 448)     # all-taps kernel: >= 48 -> 48, >= 3000 px
    assert not pref(3, 512, 512, 24, 32) and not pref(3, 512, 512, 12, 16)              # deep small maps: Winograd form
    assert pref(3, 32, 32, 384, 512) and not pref(3, 32, 32, 96, 128)                   # wide shallow layers on the tap-split kernel
    assert pref(3, 512, 512, 12, 16, direct=True)                                       # a plugin that asks for the direct form gets it
    monkeypatch.setattr(hip_ops, 'CONVK_WGRAD3_RING', False)      # the A/B switch is a module attribute (no environment variable is read)
    assert not pref(3, 128, 128, 96, 128) and pref(3, 192, 192, 96, 160)
    monkeypatch.setattr(hip_ops, 'CONVK_WGRAD3_RING', True)
    lib = _hip.lib()
    assert lib.savfi_convk_wgrad_sums_bias(8, 4, 128, 128, 96, 128, 3, 1) == 1
    assert lib.savfi_convk_wgrad_sums_bias(8, 4, 32, 32, 384, 512, 3, 1) == 0 and lib.savfi_convk_wgrad_sums_bias(2, 1, 64, 128, 128, 128, 5, 2) == 0
    assert lib.savfi_convk_wgrad_workspace_floats(8, 4, 128, 128, 96, 128, 3, 1) >= 4 * 128 * (128 * 9 + 1)


def test_the_shipped_product_reads_no_experiment_switches():
    """A/B switches are variant builds (tools/build_variant.sh: compile-time macros) or module attributes a tool sets -- not environment
    variables of the shipped product: no getenv in csrc/, and the package reads only these documented SAVFI_* variables."""
    import re
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "meta-interpolation_amd")
    allowed = {"SAVFI_HIP_LIB", "SAVFI_DIST_BACKEND", "SAVFI_PIN_DEVICE", "SAVFI_BUILD_CACHE"}
    strays = []
    for dirpath, _, files in os.walk(pkg):
        if '__pycache__' in dirpath or os.path.basename(dirpath) == 'lib':
            continue
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith(('.hip', '.h')):
                for i, line in enumerate(open(path), 1):
                    if 'getenv' in line and not line.lstrip().startswith('//'):
                        strays.append('%s:%d getenv' % (path, i))
            elif f.endswith('.py'):
                for i, line in enumerate(open(path), 1):
                    if 'environ' in line:
                        for name in re.findall(r"SAVFI_[A-Z0-9_]+", line):
                            if name not in allowed:
                                strays.append('%s:%d %s' % (path, i, name))
    assert not strays, strays
