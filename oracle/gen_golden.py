"""oracle/gen_golden.py -- TEST INFRASTRUCTURE.  Generates tests/golden/*.npz by IMPORTING THE REFERENCE.

Run in the build container only (it needs /root/reference, which never travels to the GPU box):

    python oracle/gen_golden.py            # writes tests/golden/*.npz (a few MB)

No reference file is modified or copied.  The import needs third-party shims only:
  (1) sys.modules stubs for cupy (cupy.util.memoize pass-through), torchvision (+transforms, models)
      and cv2 -- none of them does arithmetic on this path;
  (2) ReduceLROnPlateau wrapped to swallow the removed `verbose=` kwarg;
  (3) `--num_gpu 0 --resume` from a scratch cwd holding checkpoint/<exp>/checkpoint.pth so that no
      pretrained_models/*.pth is needed; the weights are the seeded recipe of
      meta-interpolation_amd/synthetic.py;
  (4) SepConv only: the reference op has no CPU branch (sepconv.py:293-294), so
      sepconv.sepconv_op.sepconv.FunctionSepconv is replaced by oracle.torch_ops.SepconvCPU;
  (5) VoxelFlow only: torch.Tensor.cuda = identity for the hard-coded .cuda() (voxel_flow.py:476-477).
Harness-level observation (no reference edit): optimizer.step and inner_loop_optimizer.update_params
are wrapped to record outer-gradient and fast-weight fingerprints.
"""
import argparse
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("SAVFI_REFERENCE", "/root/reference")
GOLD = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from meta_interpolation_amd import synthetic  # noqa: E402  (the build's own seeded recipe)
from oracle import torch_ops as O  # noqa: E402


def fingerprint(t):
    t = t.detach().double().reshape(-1)
    return np.array([t.sum().item(), t.abs().sum().item()] + t[:4].tolist() + [0.0] * max(0, 4 - t.numel()))


def install_shims():
    cupy = types.ModuleType("cupy")
    cupy.util = types.SimpleNamespace(memoize=lambda **kw: (lambda fn: fn))
    cupy.cuda = types.SimpleNamespace()
    sys.modules["cupy"] = cupy
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")

    class Normalize:  # torchvision.transforms.Normalize: (x - mean[:, None, None]) / std[:, None, None]
        def __init__(self, mean, std):
            self.mean = torch.tensor(mean, dtype=torch.float32)[:, None, None]
            self.std = torch.tensor(std, dtype=torch.float32)[:, None, None]

        def __call__(self, x):
            return (x - self.mean) / self.std
    tv.transforms.Normalize = Normalize
    tv.models = types.ModuleType("torchvision.models")
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tv.transforms
    sys.modules["torchvision.models"] = tv.models
    sys.modules["cv2"] = types.ModuleType("cv2")

    import torch.optim.lr_scheduler as sched
    orig = sched.ReduceLROnPlateau

    class Plateau(orig):
        def __init__(self, *a, verbose=None, **kw):
            super().__init__(*a, **kw)
    sched.ReduceLROnPlateau = Plateau
    torch.optim.lr_scheduler.ReduceLROnPlateau = Plateau
    sys.path.insert(0, REF)


def reference_args(**over):
    import config as ref_config
    argv = sys.argv
    sys.argv = [argv[0]]
    try:
        args, _ = ref_config.get_args()
    finally:
        sys.argv = argv
    args.num_gpu = 0
    args.cuda = False
    args.resume = True
    args.exp_name = "golden"
    for k, v in over.items():
        setattr(args, k, v)
    return args


def build_reference_system(args, model, seed=12345, recipe=None):
    """Construct the reference SceneAdaptiveInterpolation with seeded weights via the --resume route."""
    import meta_learning_system as ref_mls
    if model == 'sepconv':
        import sepconv.sepconv_op.sepconv as ref_op
        ref_op.FunctionSepconv = O.SepconvCPU
    if model in ('voxelflow', 'rrin'):       # hard-coded .cuda() calls (voxel_flow.py:476-477, rrin/model.py:11-12)
        torch.Tensor.cuda = lambda self, *a, **k: self
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp(prefix="savfi_golden_")
    os.chdir(tmp)
    try:
        # two-pass: build once with an empty checkpoint to learn the net's state_dict, then reload seeded
        os.makedirs(os.path.join("checkpoint", args.exp_name), exist_ok=True)
        for fname in ("checkpoint.pth", "model_best.pth"):      # val/test modes resume from model_best.pth (utils.py:37-40)
            torch.save({'epoch': 0, 'state_dict': {}}, os.path.join("checkpoint", args.exp_name, fname))
        system = ref_mls.SceneAdaptiveInterpolation(args)
        sd = synthetic.seeded_state_dict(system.net, model, seed, recipe)
        system.net.load_state_dict(sd)
    finally:
        os.chdir(cwd)
    return system


def observe(system, rec):
    """Wrap update_params / optimizer.step to record fingerprints (harness-side, reference untouched)."""
    rule = system.inner_loop_optimizer
    orig_update = rule.update_params

    def update_params(names_weights_dict, names_grads_wrt_params_dict, num_step, **kw):
        out = orig_update(names_weights_dict=names_weights_dict,
                          names_grads_wrt_params_dict=names_grads_wrt_params_dict, num_step=num_step, **kw)
        rec['n_live'].append(len(out))
        rec['grad_fp'].append({k: fingerprint(v) for k, v in names_grads_wrt_params_dict.items() if v is not None})
        rec['weight_fp'].append({k: fingerprint(v) for k, v in out.items()})
        return out
    rule.update_params = update_params
    orig_step = system.optimizer.step

    def step(*a, **k):
        rec['outer_grad_fp'] = {n: fingerprint(p.grad) for n, p in system.named_parameters()
                                if p.requires_grad and p.grad is not None}
        return None  # do not move the weights: the fixture describes ONE iteration from the seeded theta
    system.optimizer.step = step


def pack_fp(prefix, list_of_dicts, out):
    for i, d in enumerate(list_of_dicts):
        keys = sorted(d)
        out['%s_%d_keys' % (prefix, i)] = np.array(keys)
        out['%s_%d' % (prefix, i)] = np.stack([d[k] for k in keys]) if keys else np.zeros((0, 6))


SYSTEM_CASES = {
    # name: (model, H, W, tasks, args overrides)
    'c1_cain_lslr_sgd': ('cain', 64, 64, 1, dict(optimizer='SGD', inner_lr=1e-3, number_of_training_steps_per_iter=1,
                                                 number_of_evaluation_steps_per_iter=1, loss='1*L1')),
    'cain_l2f': ('cain', 64, 64, 1, dict(optimizer='SGD', inner_lr=1e-3, attenuate=True, loss='1*L1')),
    'sepconv_lslr_sgd_2step': ('sepconv', 64, 64, 1, dict(optimizer='SGD', inner_lr=1e-3, loss='1*L1',
                                                          number_of_training_steps_per_iter=2,
                                                          number_of_evaluation_steps_per_iter=2)),
    'sepconv_metasgd_adamax_2step': ('sepconv', 64, 64, 1, dict(optimizer='Adamax', inner_lr=1e-4, metasgd=True,
                                                                loss='1*L1',
                                                                number_of_training_steps_per_iter=2,
                                                                number_of_evaluation_steps_per_iter=2)),
    'sepconv_msl_learnable_2step': ('sepconv', 64, 64, 2, dict(optimizer='SGD', inner_lr=1e-3, loss='1*L1',
                                                               number_of_training_steps_per_iter=2,
                                                               number_of_evaluation_steps_per_iter=2,
                                                               use_multi_step_loss_optimization=True,
                                                               multi_step_loss_num_epochs=10,
                                                               learnable_per_layer_per_step_inner_loop_learning_rate=True)),
    'voxelflow_metasgd_adamax_2step': ('voxelflow', 64, 64, 2, dict(optimizer='Adamax', inner_lr=1e-4, metasgd=True,
                                                                    loss='1*MSE',
                                                                    number_of_training_steps_per_iter=2,
                                                                    number_of_evaluation_steps_per_iter=2)),
    'cain_lslr_adam_1step': ('cain', 64, 64, 1, dict(optimizer='Adam', inner_lr=1e-4, loss='1*L1')),
    # smooth-rule VoxelFlow case (tight parity gate; the Adamax case above is chaotic by construction)
    'voxelflow_lslr_sgd_2step': ('voxelflow', 64, 64, 2, dict(optimizer='SGD', inner_lr=1e-3, loss='1*MSE',
                                                              number_of_training_steps_per_iter=2,
                                                              number_of_evaluation_steps_per_iter=2)),
    # the two plugins built on the pixel-flow warp
    'rrin_lslr_sgd_2step': ('rrin', 64, 64, 1, dict(optimizer='SGD', inner_lr=1e-3, loss='1*L1',
                                                    number_of_training_steps_per_iter=2,
                                                    number_of_evaluation_steps_per_iter=2)),
    'superslomo_lslr_sgd_2step': ('superslomo', 64, 64, 2, dict(optimizer='SGD', inner_lr=1e-3, loss='1*L1',
                                                                number_of_training_steps_per_iter=2,
                                                                number_of_evaluation_steps_per_iter=2)),
    # 128 x 128 twins of two 64 x 64 cases: deepest maps 4 x 4 instead of 2 x 2, so that the OUTER-gradient fingerprints are not decided
    # by one ReLU unit of a 2 x 2 map switching under another summation order (gated at the plain 1e-3)
    'sepconv_msl_learnable_2step_128': ('sepconv', 128, 128, 2, dict(optimizer='SGD', inner_lr=1e-3, loss='1*L1',
                                                                   number_of_training_steps_per_iter=2,
                                                                   number_of_evaluation_steps_per_iter=2,
                                                                   use_multi_step_loss_optimization=True,
                                                                   multi_step_loss_num_epochs=10,
                                                                   learnable_per_layer_per_step_inner_loop_learning_rate=True)),
    'superslomo_lslr_sgd_2step_128': ('superslomo', 128, 128, 2, dict(optimizer='SGD', inner_lr=1e-3, loss='1*L1',
                                                                    number_of_training_steps_per_iter=2,
                                                                    number_of_evaluation_steps_per_iter=2)),
    # the reference's own launch-script settings: run_voxelflow.sh / run_cain.sh (Adam + Meta-SGD, 1 step, lr 1e-5)
    'voxelflow_script_metasgd_adam_1step': ('voxelflow', 64, 64, 1, dict(optimizer='Adam', inner_lr=1e-5, metasgd=True,
                                                                         loss='1*MSE')),
}


def run_system_case(name):
    model, H, W, B, over = SYSTEM_CASES[name]
    args = reference_args(model=model, batch_size=B, **over)
    out = {'model': np.array(model), 'H': H, 'W': W, 'B': B,
           'args': np.array(repr(sorted(over.items())))}
    frames = synthetic.septuplet_batch(B, H, W, model=model)
    for phase in ('train', 'val'):
        torch.manual_seed(0)
        system = build_reference_system(args, model)
        if getattr(args, 'attenuate', False):
            # gamma_mult is initialised to 0 (meta_learning_system.py:117) which would make L2F a no-op;
            # give the attenuator a seeded, non-trivial state so the fixture exercises it.
            rs = np.random.RandomState(777)
            with torch.no_grad():
                system.gamma_mult.fill_(0.5)
                for p in system.attenuator.parameters():
                    p.copy_(torch.from_numpy(rs.uniform(-0.05, 0.05, size=tuple(p.shape)).astype(np.float32)))
        rec = dict(n_live=[], grad_fp=[], weight_fp=[], outer_grad_fp={})
        observe(system, rec)
        if phase == 'train':
            losses, preds, metrics = system.run_train_iter(data_batch=[f.clone() for f in frames], epoch=0,
                                                           do_evaluation=True)
        else:
            losses, preds, metrics = system.run_validation_iter(data_batch=[f.clone() for f in frames])
        out[phase + '_loss'] = np.float64(losses['loss'].item())
        for k, v in losses.items():
            if k != 'loss' and not k.startswith('loss_importance'):
                out[phase + '_part_' + k] = np.float64(v)
        out[phase + '_preds'] = torch.stack([p.squeeze(0) for p in preds]).numpy()
        out[phase + '_psnr'] = np.float64(metrics['psnr'].avg)
        out[phase + '_ssim'] = np.float64(float(metrics['ssim'].avg))
        out[phase + '_n_live'] = np.array(rec['n_live'])
        pack_fp(phase + '_grad_fp', rec['grad_fp'], out)
        pack_fp(phase + '_weight_fp', rec['weight_fp'], out)
        if phase == 'train':
            pack_fp('outer_grad_fp', [rec['outer_grad_fp']], out)
        print('  %-34s %-5s loss=%.8f psnr=%.4f n_live=%s' % (name, phase, out[phase + '_loss'],
                                                              out[phase + '_psnr'], rec['n_live']), flush=True)
    np.savez_compressed(os.path.join(GOLD, 'system_%s.npz' % name), **out)


def run_test_mode_case():
    """run_test_iter (meta_learning_system.py:630-697): adapt on a 4-frame clip, interpolate between frames 1 and 2."""
    out = {}
    for model, over in (('sepconv', dict(optimizer='SGD', inner_lr=1e-3, loss='1*L1', number_of_evaluation_steps_per_iter=2)),
                        ('cain', dict(optimizer='SGD', inner_lr=1e-3, loss='1*L1', number_of_evaluation_steps_per_iter=1)),
                        ('rrin', dict(optimizer='SGD', inner_lr=1e-3, loss='1*L1', number_of_evaluation_steps_per_iter=1)),
                        ('superslomo', dict(optimizer='SGD', inner_lr=1e-3, loss='1*L1', number_of_evaluation_steps_per_iter=1))):
        args = reference_args(model=model, batch_size=1, mode='test', **over)
        system = build_reference_system(args, model)
        frames = synthetic.septuplet_batch(2, 64, 64, model=model, frames=4)
        preds = system.run_test_iter(data_batch=[f.clone() for f in frames])
        out[model + '_preds'] = torch.stack(preds).numpy()
        out[model + '_args'] = np.array(repr(sorted(over.items())))
        print('  test-mode %s: %s mean %.6f' % (model, tuple(out[model + '_preds'].shape), out[model + '_preds'].mean()))
    np.savez_compressed(os.path.join(GOLD, 'test_mode.npz'), **out)


def run_rule_cases():
    """Each reference rule x optimizer for tau = 1..3 on seeded tensors (one tensor gets a None grad from
    step 2 on, like SepConv's subnets)."""
    import inner_loop_optimizers as ref_rules
    out = {}
    rs = np.random.RandomState(42)
    shapes = {'a.weight': (4, 3, 3, 3), 'a.bias': (4,), 'b.weight': (5, 7), 'c.weight': (129,)}
    w0 = {k: torch.from_numpy(rs.normal(size=s).astype(np.float32)) for k, s in shapes.items()}
    grads = [{k: torch.from_numpy(rs.normal(size=s).astype(np.float32)) for k, s in shapes.items()} for _ in range(3)]
    for k in w0:
        out['w0/' + k] = w0[k].numpy()
        for t in range(3):
            out['g%d/%s' % (t, k)] = grads[t][k].numpy()
    for kind in ('lslr', 'metasgd'):
        for opt in ('SGD', 'Adam', 'Adamax'):
            if kind == 'lslr':
                rule = ref_rules.LSLRGradientDescentLearningRule(device=torch.device('cpu'), optimizer=opt,
                                                                 total_num_inner_loop_steps=3,
                                                                 use_learnable_learning_rates=False,
                                                                 init_learning_rate=0.01)
            else:
                rule = ref_rules.MetaSGDLearningRule(device=torch.device('cpu'), optimizer=opt,
                                                     init_learning_rate=0.01)
            rule.initialize(w0)
            with torch.no_grad():  # non-uniform learning rates
                for i, (k, p) in enumerate(rule.names_learning_rates_dict.items()):
                    p.mul_(torch.from_numpy(np.random.RandomState(100 + i).uniform(0.5, 1.5, size=tuple(p.shape)).astype(np.float32)))
                    out['lr/%s/%s' % (kind, k)] = p.detach().numpy().copy()
            rule.initialize_state()
            w = dict(w0)
            for t in range(3):
                g = dict(grads[t])
                if t >= 1:
                    g['c.weight'] = None
                    if kind == 'metasgd' and opt == 'SGD':
                        del g['c.weight']   # the reference raises TypeError on None here (:328-330)
                g = {k: g[k] for k in w if k in g}
                with torch.no_grad():
                    w = rule.update_params(w, g, t)
                for k, v in w.items():
                    out['out/%s/%s/%d/%s' % (kind, opt, t, k)] = v.numpy().copy()
    np.savez_compressed(os.path.join(GOLD, 'rules.npz'), **out)
    print('  rules.npz: %d arrays' % len(out))


def run_op_cases():
    import model_utils as ref_mu
    import utils as ref_utils
    out = {}
    rs = np.random.RandomState(7)
    x = torch.from_numpy(rs.normal(size=(2, 3, 16, 24)).astype(np.float32))
    out['ps_in'] = x.numpy()
    out['ps_down8'] = ref_mu.pixel_shuffle(x, 1 / 8).numpy()
    out['ps_down2'] = ref_mu.pixel_shuffle(x, 1 / 2).numpy()
    y = torch.from_numpy(rs.normal(size=(1, 192, 2, 3)).astype(np.float32))
    out['ps_up_in'] = y.numpy()
    out['ps_up8'] = ref_mu.pixel_shuffle(y, 8).numpy()
    # PSNR / SSIM (the parity metric)
    a = torch.from_numpy(rs.uniform(size=(3, 40, 56)).astype(np.float32))
    b = (a + torch.from_numpy(rs.normal(scale=0.05, size=(3, 40, 56)).astype(np.float32))).clamp(0, 1)
    psnr, ssim = ref_utils.calc_metrics(a, b)
    out['metric_a'], out['metric_b'] = a.numpy(), b.numpy()
    out['metric_psnr'], out['metric_ssim'] = np.float64(psnr), np.float64(float(ssim))
    # MSL importance vectors
    import meta_learning_system as ref_mls
    for S, epoch, E in [(5, 0, 10), (5, 3, 10), (5, 50, 10), (1, 0, 1), (3, 2, 4)]:
        stub = types.SimpleNamespace(args=types.SimpleNamespace(number_of_training_steps_per_iter=S,
                                                                multi_step_loss_num_epochs=E),
                                     current_epoch=epoch, device=torch.device('cpu'))
        v = ref_mls.SceneAdaptiveInterpolation.get_per_step_loss_importance_vector(stub)
        out['msl_%d_%d_%d' % (S, epoch, E)] = v.numpy()
    # VoxelFlow warp: whole-model forward on seeded weights, plus the captured tanh map
    torch.Tensor.cuda = lambda self, *a, **k: self
    from voxelflow.core.models.voxel_flow import MetaVoxelFlow
    net = MetaVoxelFlow(types.SimpleNamespace(), resume=False)
    net.load_state_dict(synthetic.seeded_state_dict(net, 'voxelflow'))
    fr = synthetic.septuplet_batch(1, 48, 80, model='voxelflow')
    captured = {}
    hook = net.conv4.register_forward_hook(lambda m, i, o: captured.setdefault('pre_tanh', o.detach().clone()))
    with torch.no_grad():
        o = net(fr[0], fr[2])
    hook.remove()
    out['vf_f0'], out['vf_f1'], out['vf_out'] = fr[0].numpy(), fr[2].numpy(), o.numpy()
    out['vf_x3'] = torch.tanh(captured['pre_tanh']).numpy()
    # an aggressive synthetic flow (samples leave the frame) through the reference's warp tail alone:
    # replay forward() with conv4 replaced by a constant map
    big = torch.tanh(torch.from_numpy(rs.normal(scale=1.5, size=(1, 3, 64, 128)).astype(np.float32)))
    net.conv4.register_forward_hook(lambda m, i, o: torch.atanh(big.clamp(-0.9999, 0.9999)))
    with torch.no_grad():
        o2 = net(fr[0], fr[2])
    out['vf_big_x3'] = torch.tanh(torch.atanh(big.clamp(-0.9999, 0.9999))).numpy()
    out['vf_big_out'] = o2.numpy()
    # pixel-flow backward warp: the reference's own backWarp (superslomo/model.py:231-307) and warp (rrin/model.py:8-20)
    # on a seeded image and a flow whose targets partly leave the frame; flow gradient by autograd
    from superslomo.model import backWarp
    from rrin.model import warp as rrin_warp
    rw = np.random.RandomState(11)
    img = torch.from_numpy(rw.uniform(size=(2, 3, 20, 36)).astype(np.float32))
    flow = torch.from_numpy(rw.normal(scale=4.0, size=(2, 2, 20, 36)).astype(np.float32)).requires_grad_()
    gout = torch.from_numpy(rw.normal(size=(2, 3, 20, 36)).astype(np.float32))
    warped = backWarp(36, 20, torch.device('cpu'))(img, flow)
    gflow, = torch.autograd.grad((warped * gout).sum(), flow)
    out['fw_img'], out['fw_flow'], out['fw_gout'] = img.numpy(), flow.detach().numpy(), gout.numpy()
    out['fw_out'], out['fw_gflow'] = warped.detach().numpy(), gflow.numpy()
    out['fw_out_rrin'] = rrin_warp(img, flow.detach()).numpy()
    np.savez_compressed(os.path.join(GOLD, 'ops.npz'), **out)
    print('  ops.npz: %d arrays' % len(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', nargs='*', default=None)
    opts = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    install_shims()
    todo = opts.only or (['rules', 'ops', 'test_mode'] + list(SYSTEM_CASES))
    for item in todo:
        print('[golden]', item, flush=True)
        if item == 'rules':
            run_rule_cases()
        elif item == 'ops':
            run_op_cases()
        elif item == 'test_mode':
            run_test_mode_case()
        else:
            run_system_case(item)


if __name__ == '__main__':
    main()
