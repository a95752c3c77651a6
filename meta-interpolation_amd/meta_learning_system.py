"""SceneAdaptiveInterpolation: the MAML / MAML++ / Meta-SGD / L2F system around a VFI plugin.

Drop-in for the reference class of the same name (meta_learning_system.py:29-697): same constructor
(``SceneAdaptiveInterpolation(args)``), same methods called by ExperimentBuilder
(``run_train_iter / run_validation_iter / run_test_iter``), same attributes (``optimizer``,
``scheduler``, ``net``, ``inner_loop_optimizer``, ``attenuator``, ``gamma_mult``, ``mean``/``std``) and
the same ``state_dict`` key layout.  Per task and inner step: two support passes ((0,4)->2 and
(2,6)->4), one ``autograd.grad`` over both, one per-parameter update; then the target pass (2,4)->3.

What is different (MI355X-first, results identical):
  * model ops and the update are HIP kernels behind the C ABI (sepconv, voxel warp, pixel shuffle,
    fused multi-tensor update, L2F mean/scale, fused L1/MSE);
  * the two support triplets of a step -- same weights, independent samples -- run as one N=2 forward
    (``--fuse_support_pairs 0`` restores two N=1 calls);
  * no host synchronisation inside the task loop: the reference pulls every loss to the host
    (:332) and every ``zero_grad(params)`` syncs per parameter; here loss scalars stay on the device
    and are fetched once per iteration;
  * tasks shard across processes (task_parallel.TaskParallel) with one all-reduce of outer grads;
    with a single process this is the reference's sequential loop.  The reference's ``.module`` call
    for >1 visible device (:285-287) is not reproduced: one process owns one device.
"""
import functools
import os
import threading

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from . import _hip, graph_inner_loop, hip_ops, model_utils, utils
from .inner_loop_optimizers import LSLRGradientDescentLearningRule, MetaSGDLearningRule
from .loss import Loss
from .task_parallel import TaskParallel


def set_torch_seed(seed):
    """Seed torch from a numpy stream, returning that stream (reference :16-26)."""
    rng = np.random.RandomState(seed=seed)
    torch.manual_seed(seed=int(rng.randint(0, 999999)))
    return rng


# ---------------------------------------------------------------------------------------------
# --model plugin surface
# ---------------------------------------------------------------------------------------------
def _build_sepconv(args, resume):
    from .sepconv.model import MetaNetwork
    return MetaNetwork(resume=resume, strModel='l1', windowed=bool(getattr(args, 'sepconv_window', 1)))


def _build_cain(args, resume):
    from .cain.model import MetaCAIN
    return MetaCAIN(depth=3, resume=resume)


def _build_voxelflow(args, resume):
    from .voxelflow.core.models.voxel_flow import MetaVoxelFlow
    return MetaVoxelFlow(args, resume=resume)


def _build_rrin(args, resume):
    from .rrin.model import MetaRRIN
    return MetaRRIN(level=3, resume=resume)


def _build_superslomo(args, resume):
    from .superslomo.model import MetaSuperSloMo
    return MetaSuperSloMo(torch.device('cuda') if args.cuda else torch.device('cpu'), resume=resume)


MODEL_REGISTRY = {'sepconv': _build_sepconv, 'cain': _build_cain, 'voxelflow': _build_voxelflow, 'rrin': _build_rrin,
                  'superslomo': _build_superslomo}


def register_model(name, factory):
    """Add a ``--model`` plugin: ``factory(args, resume) -> nn.Module`` with
    ``forward(frame0, frame1, params=None, **kw)``, ``zero_grad(params)``, ``restore_backup_stats()``."""
    MODEL_REGISTRY[name] = factory


class _DeferredMeters:
    """Collects device scalars during the task loop; turns them into AverageMeters with ONE sync."""

    def __init__(self):
        self.items = {}

    def add(self, key, value):
        self.items.setdefault(key, []).append(value.detach())

    def flush(self):
        meters = {}
        if not self.items:
            return meters
        keys = list(self.items)
        flat = torch.stack([v.reshape(()) for k in keys for v in self.items[k]]).cpu().numpy()
        i = 0
        for k in keys:
            m = utils.AverageMeter()
            for _ in self.items[k]:
                m.update(flat[i])
                i += 1
            meters[k] = m
        return meters


class _LazyLog(dict):
    """The dict run_train_iter returns (`losses`, `metrics`): what needs the device to have finished -- logged losses, PSNR / SSIM
    meters, the importance vector -- is fetched on the FIRST READ instead of before returning.  A caller that logs every
    iteration (ExperimentBuilder, the reference's loop) syncs exactly where it did; one that does not read (a timing loop)
    lets the host run on into the next iteration while the device finishes this one -- the one host sync per iteration was a
    1-5 ms bubble at every iteration boundary (profiles/r03_*_one_iteration.txt: the largest gap of each trace)."""

    def __init__(self, data, ensure):
        super().__init__(data)
        self._data, self._ensure = data, ensure

    def _ready(self):
        ensure, self._ensure = self._ensure, None
        if ensure is not None:
            ensure()
            dict.update(self, self._data)

    def __reduce__(self):            # pickle / torch.save / copy.deepcopy: a plain dict of the finished values
        self._ready()
        return (dict, (dict(self),))


def _lazy_forward(name):
    plain = getattr(dict, name)

    def method(self, *a, **kw):
        self._ready()
        return plain(self, *a, **kw)
    method.__name__ = name
    return method


# every accessor and mutator of dict goes through _ready() first: a partly overridden dict subclass would let copy(), pop(),
# ==, | ... see the dict as it was before the device finished
for _name in ('__getitem__', '__setitem__', '__delitem__', '__contains__', '__iter__', '__len__', '__repr__', '__eq__', '__ne__',
              '__or__', '__ror__', '__ior__', '__reversed__', 'get', 'keys', 'values', 'items', 'copy', 'pop', 'popitem',
              'setdefault', 'update', 'clear'):
    setattr(_LazyLog, _name, _lazy_forward(_name))
del _name


class SceneAdaptiveInterpolation(nn.Module):
    MAX_GRAPH_SETS = 8

    def __init__(self, args, net=None, inner_loop_optimizer=None, criterion=None, task_parallel=None):
        super().__init__()
        self.args = args
        self.device = torch.device('cuda') if args.cuda else torch.device('cpu')
        self.batch_size = args.batch_size
        self.use_cuda = args.cuda
        self.current_epoch = 0
        self.fuse_support_pairs = bool(getattr(args, 'fuse_support_pairs', 1))

        # 7-frame septuplets: supports (0,4)->2 and (2,6)->4, target (2,4)->3; 4-frame clips at test time
        self.support_idxs = [[0, 1, 2], [1, 2, 3]] if args.mode == 'test' else [[0, 2, 4], [2, 4, 6]]
        self.target_idxs = [2, 3, 4]

        self.rng = set_torch_seed(seed=args.random_seed)
        if net is not None:
            self.net = net.to(self.device)
        else:
            if args.model not in MODEL_REGISTRY:
                raise NotImplementedError('Model not implemented yet!')
            print('Building %s model...' % args.model)
            self.net = MODEL_REGISTRY[args.model](args, not args.resume).to(self.device)
        if args.model == 'voxelflow':
            half = torch.full((3, 1, 1), 0.5 * 255, device=self.device)
            self.mean, self.std = half.clone(), half.clone()
        if args.model == 'superslomo':
            # Super SloMo frames are mean-subtracted by the loaders; revNormalize brings outputs back to 0..1
            # (the reference builds transforms.Normalize(mean=-m, std=1) for this, :69-73)
            shift = torch.tensor([.429, .431, .397], device=self.device).view(3, 1, 1)
            self.revNormalize = lambda img: img + shift

        self.inner_learning_rate = args.inner_lr
        if inner_loop_optimizer is not None:
            self.inner_loop_optimizer = inner_loop_optimizer
        elif args.metasgd:
            print('Adaptation with Meta-SGD')
            self.inner_loop_optimizer = MetaSGDLearningRule(device=self.device, optimizer=args.optimizer,
                                                            init_learning_rate=self.inner_learning_rate)
        else:
            self.inner_loop_optimizer = LSLRGradientDescentLearningRule(
                device=self.device, optimizer=args.optimizer, init_learning_rate=self.inner_learning_rate,
                total_num_inner_loop_steps=args.number_of_training_steps_per_iter,
                use_learnable_learning_rates=args.learnable_per_layer_per_step_inner_loop_learning_rate)

        names_weights = self.get_inner_loop_parameter_dict(self.net.named_parameters())
        self.inner_loop_optimizer.initialize(names_weights_dict=names_weights)

        if args.attenuate:  # L2F: gamma = 1 - gamma_mult * MLP(layer-wise mean grads)
            L = len(names_weights)
            print('# of layers: %d' % L)
            self.attenuator = nn.Sequential(nn.Linear(L, L), nn.ReLU(inplace=True), nn.Linear(L, L),
                                            nn.Sigmoid()).to(self.device)
            self.gamma_mult = nn.Parameter(torch.zeros(1))

        self.to(self.device)

        if args.optimizer == 'Adam':
            if args.model == 'voxelflow' and hasattr(self.net, 'get_optim_policies'):
                self.optimizer = optim.Adam(self.net.get_optim_policies(), lr=args.outer_lr,
                                            weight_decay=args.weight_decay)
            else:
                self.optimizer = optim.Adam(self.trainable_parameters(), lr=args.outer_lr, betas=(0.9, 0.99))
        elif args.optimizer == 'Adamax':
            self.optimizer = optim.Adamax(self.trainable_parameters(), lr=args.outer_lr, betas=(0.9, 0.999))
        else:
            self.optimizer = optim.SGD(self.trainable_parameters(), lr=args.outer_lr)
        self.scheduler = optim.lr_scheduler.ReduceLROnPlateau(optimizer=self.optimizer, mode='min', factor=0.2,
                                                              patience=5)
        print('# of parameters: %d' % sum(p.numel() for p in self.trainable_parameters()))

        self.criterion = criterion if criterion is not None else Loss(args)
        self.task_parallel = task_parallel if task_parallel is not None else TaskParallel()
        if self.device.type == 'cuda':
            with torch.cuda.device(self.device):
                _hip.ws_watch()
        self._first_order = False
        self._defer_logging = False
        self._pending_logging = None
        self._filter_modules = None
        self._task_stream_pool = None
        self._routes = None          # (routed, unrouted) inner-loop tensor names, probed once (_routing)
        self._graphs = {}            # (frame shape, steps, training, msl) -> GraphedInnerLoop
        self._manual_grads = None    # OuterGradAccumulator of the last graphed training forward

        if args.resume:
            print('Resume training')
            utils.load_checkpoint(args, self, None)
        if args.pretrained_model is not None:
            print('Loading pretrained model: %s' % args.pretrained_model)
            ckpt = torch.load(args.pretrained_model, map_location='cpu', weights_only=False)
            with torch.no_grad():
                if 'state_dictFC' in ckpt:       # Super SloMo's released checkpoints (reference :162-167)
                    self.net.flowComp.load_state_dict(ckpt['state_dictFC'])
                    self.net.arbTimeFlowIntrp.load_state_dict(ckpt['state_dictAT'])
                else:
                    utils.lossy_load_state_dict(self.net, ckpt['state_dict'])

    # -----------------------------------------------------------------------------------------
    # small pieces kept with the reference's names
    # -----------------------------------------------------------------------------------------
    def get_per_step_loss_importance_vector(self):
        """MAML++ multi-step-loss weights (reference :186-210): uniform 1/S, non-final entries decay
        with the epoch down to 0.03/S, the final one grows up to 1-(S-1)*0.03/S."""
        S = self.args.number_of_training_steps_per_iter
        # the vector only depends on the epoch: keep the device copy.  A blocking host-to-device copy of a pageable array makes the
        # host wait for everything queued on the stream -- once per forward that was a full host/GPU sync at the top of every
        # meta-iteration (11.7 ms of a 23 ms iteration of config C1 under cProfile: the host could never run ahead of the replays)
        key = (S, self.current_epoch, self.args.multi_step_loss_num_epochs, str(self.device))
        cached = getattr(self, '_importance_cache', None)
        if cached is not None and cached[0] == key:
            return cached[1]
        if S == 0:
            vec = torch.ones(1, device=self.device)
            self._importance_cache = (key, vec)
            return vec
        w = np.ones(shape=(S,)) * (1.0 / S)
        decay = 1.0 / S / self.args.multi_step_loss_num_epochs
        floor = 0.03 / S
        for i in range(S - 1):
            w[i] = np.maximum(w[i] - self.current_epoch * decay, floor)
        w[-1] = np.minimum(w[-1] + self.current_epoch * (S - 1) * decay, 1.0 - (S - 1) * floor)
        vec = torch.Tensor(w).to(device=self.device)
        self._importance_cache = (key, vec)
        return vec

    def get_inner_loop_parameter_dict(self, params):
        """{name: param} of what the inner loop adapts (reference :213-228)."""
        keep_bn = self.args.enable_inner_loop_optimizable_bn_params
        return {name: p for name, p in params if p.requires_grad and (keep_bn or "norm_layer" not in name)}

    def trainable_parameters(self):
        for p in self.parameters():
            if p.requires_grad:
                yield p

    # -----------------------------------------------------------------------------------------
    # forward passes
    # -----------------------------------------------------------------------------------------
    def net_forward(self, frame0, frame1, target, weights, backup_running_statistics, training, num_step):
        """One backbone pass with fast weights + criterion -> (losses dict, output)  (reference :475-509)."""
        output = self.net.forward(frame0, frame1, params=weights,
                                  backup_running_statistics=backup_running_statistics, num_step=num_step)
        if isinstance(output, tuple):     # superslomo: (frame, flows and warped frames for the 'Super' loss)  (:499-502)
            output, extras = output
            return self.criterion(output, target, I0=frame0, I1=frame1, **extras), output
        return self.criterion(output, target), output

    def _support_loss(self, frames, task_id, weights, num_step):
        """Sum of the two support-triplet losses of one inner step (reference :387-396)."""
        a, b = self.support_idxs
        # step 0 differentiates w.r.t. theta itself, whose non-routed tensors ARE the modules' own parameters
        # (reference fact: 94 live tensors at step 0, 54 afterwards); from step 1 on those are constants
        model_utils.set_own_params_const(self._first_order and num_step > 0)
        # first-order support pass: its weight gradients may run beside the data-gradient chain (joined by the callers
        # right after autograd.grad, before anything reads them)
        hip_ops.set_weight_gradient_overlap(self._first_order and self.device.type == 'cuda'
                                            and bool(getattr(self.args, 'wgrad_overlap', 0)))
        try:
            return self._support_loss_impl(frames, task_id, weights, num_step, a, b)
        finally:
            model_utils.set_own_params_const(False)
            hip_ops.set_weight_gradient_overlap(False)

    def _support_loss_impl(self, frames, task_id, weights, num_step, a, b):
        if self.fuse_support_pairs:
            sl = slice(task_id, task_id + 1)
            f0 = torch.cat([frames[a[0]][sl], frames[b[0]][sl]], 0)
            f1 = torch.cat([frames[a[2]][sl], frames[b[2]][sl]], 0)
            out = self.net.forward(f0, f1, params=weights, backup_running_statistics=(num_step == 0),
                                   num_step=num_step)
            if isinstance(out, tuple):    # superslomo; its extras only feed the 'Super' loss, which Loss rejects
                out = out[0]
            la = self.criterion(out[0:1], frames[a[1]][sl])
            lb = self.criterion(out[1:2], frames[b[1]][sl])
            return la['total'] + lb['total']
        total = 0
        for ind in (a, b):
            losses, _ = self.net_forward(frame0=frames[ind[0]][task_id].unsqueeze(0),
                                         frame1=frames[ind[2]][task_id].unsqueeze(0),
                                         target=frames[ind[1]][task_id].unsqueeze(0), weights=weights,
                                         backup_running_statistics=(num_step == 0), training=True,
                                         num_step=num_step)
            total = total + losses['total']
        return total

    def _target_pass(self, frames, task_id, weights, num_step):
        t = self.target_idxs
        return self.net_forward(frame0=frames[t[0]][task_id].unsqueeze(0), frame1=frames[t[2]][task_id].unsqueeze(0),
                                target=frames[t[1]][task_id].unsqueeze(0), weights=weights,
                                backup_running_statistics=False, training=True, num_step=num_step)

    # -----------------------------------------------------------------------------------------
    # L2F
    # -----------------------------------------------------------------------------------------
    def get_task_embeddings(self, frames, task_id, names_weights_copy):
        """Layer-wise mean of the support gradient at theta (reference :231-255), one fused reduction."""
        loss = self._support_loss(frames, task_id, names_weights_copy, num_step=0)
        self.net.zero_grad(names_weights_copy)
        grads = torch.autograd.grad(loss, names_weights_copy.values(), create_graph=False, allow_unused=True)
        hip_ops.join_weight_gradients()
        if any(g is None for g in grads):
            raise AttributeError("L2F needs a gradient for every inner-loop tensor (the reference calls "
                                 ".mean() on None here, meta_learning_system.py:251)")
        return hip_ops.mt_mean(grads)

    def attenuate_init(self, task_embeddings, names_weights_copy):
        """w_i <- gamma_i * w_i with gamma = clamp(1 - gamma_mult * attenuator(e), 0, 1)  (reference :258-272)."""
        gamma = 1 - self.gamma_mult * self.attenuator(task_embeddings)
        gamma.clamp_(0, 1)
        scaled = hip_ops.mt_scale(gamma, list(names_weights_copy.values()))
        return dict(zip(names_weights_copy.keys(), scaled))

    # -----------------------------------------------------------------------------------------
    # inner loop
    # -----------------------------------------------------------------------------------------
    def apply_inner_loop_update(self, loss, names_weights_copy, use_second_order, current_step_idx):
        """grad of the support loss w.r.t. the fast weights, then the learned update (reference :275-321).
        Tensors without a gradient drop out of the dict."""
        self.net.zero_grad(params=names_weights_copy)
        grads = torch.autograd.grad(loss, names_weights_copy.values(), create_graph=use_second_order,
                                    allow_unused=True)
        hip_ops.join_weight_gradients()
        return self.inner_loop_optimizer.update_params(
            names_weights_dict=names_weights_copy,
            names_grads_wrt_params_dict=dict(zip(names_weights_copy.keys(), grads)),
            num_step=current_step_idx)

    def _adapt(self, frames, task_id, num_steps, use_second_order, per_step_hook=None):
        """Run the inner loop of one task; returns the adapted fast weights."""
        weights = self.get_inner_loop_parameter_dict(self.net.named_parameters())
        weights = {name.replace('module.', ''): v for name, v in weights.items()}
        self.inner_loop_optimizer.initialize_state()
        if self.args.attenuate:
            emb = self.get_task_embeddings(frames, task_id, weights)
            weights = self.attenuate_init(task_embeddings=emb, names_weights_copy=weights)
        for num_step in range(num_steps):
            support_loss = self._support_loss(frames, task_id, weights, num_step)
            weights = self.apply_inner_loop_update(loss=support_loss, names_weights_copy=weights,
                                                   use_second_order=use_second_order, current_step_idx=num_step)
            if per_step_hook is not None:
                per_step_hook(num_step, weights)
        return weights

    def _to_unit_range(self, img):
        if self.args.model == 'voxelflow':
            return (img * self.std + self.mean) / 255.0
        if self.args.model == 'superslomo':
            return self.revNormalize(img)
        return img

    def _task_body(self, frames, task_id, *, num_steps, use_second_order, msl, training_phase, do_evaluation, importance):
        """Everything one task contributes to a meta-iteration (reference :366-461): adaptation, target pass(es), its
        loss term, prediction, logging scalars.  Touches no shared mutable state, so tasks can run concurrently."""
        task_losses, logs, state = [], [], {}

        def after_step(num_step, weights):
            if msl:  # MAML++: weighted target loss after every inner step
                tl, tp_ = self._target_pass(frames, task_id, weights, num_step)
                task_losses.append(importance[num_step] * tl['total'])
                logs.extend(tl.items())
                state['preds'] = tp_

        weights = self._adapt(frames, task_id, num_steps, use_second_order, after_step)

        if not training_phase:
            with torch.no_grad():
                tl, state['preds'] = self._target_pass(frames, task_id, weights, num_steps)
            task_losses.append(tl['total'])
            logs.extend(tl.items())
        elif not msl:
            tl, state['preds'] = self._target_pass(frames, task_id, weights, num_steps)
            task_losses.append(tl['total'])
            logs.extend(tl.items())

        target_preds = state['preds']
        res = {'pred': self._to_unit_range(target_preds.detach().squeeze(0)).unsqueeze(0), 'logs': logs}
        if do_evaluation:
            out01 = self._to_unit_range(target_preds.detach().squeeze(0))
            tgt01 = self._to_unit_range(frames[self.target_idxs[1]][task_id].detach())
            q_o, q_t = utils.quantize(out01, 1.), utils.quantize(tgt01, 1.)
            res['mse'] = (q_o - q_t).div(255).pow(2).mean()
            res['ssim'] = utils.ssim(q_o.unsqueeze(0), q_t.unsqueeze(0), val_range=255)
        res['loss'] = torch.sum(torch.stack(task_losses))
        if not training_phase:
            self.net.restore_backup_stats()
        return res

    # -----------------------------------------------------------------------------------------
    # tasks in lockstep (--task_batch T): the reference's sequential task loop (:366) as ONE pass per inner step
    # -----------------------------------------------------------------------------------------
    def _routing(self, frame_shape):
        """(routed, unrouted) inner-loop tensor names: which of them the plugin actually reads from the fast-weight dict
        (SepConv 54 of 94, VoxelFlow 9 of 23, CAIN 494 of 494: SURVEY.md fact 6).  One eager pass on cloned tensors, once."""
        if self._routes is None:
            theta = self.get_inner_loop_parameter_dict(self.net.named_parameters())
            fast = {k: v.detach().clone().requires_grad_() for k, v in theta.items()}
            x = torch.zeros((1,) + tuple(frame_shape), device=self.device)
            out = self.net.forward(x, x, params=fast, backup_running_statistics=False, num_step=0)
            out = out[0] if isinstance(out, tuple) else out
            g = torch.autograd.grad(out.sum(), list(fast.values()), allow_unused=True)
            self._routes = ([k for k, gi in zip(fast, g) if gi is not None], [k for k, gi in zip(fast, g) if gi is None])
        return self._routes

    def _routing_known_incomplete(self):
        """True when the routing probe has run and found inner-loop tensors the plugin never reads from the fast dict (graphed
        L2F needs every tensor routed); before the probe has run the answer is 'not known to be incomplete'."""
        return self._routes is not None and bool(self._routes[1])

    def _lockstep_width(self, use_second_order, frame_shape):
        """Tasks per lockstep group, or 0 when this pass must take the sequential loop (second order; L2F on a plugin that
        does not route every tensor: its per-task embedding needs per-task gradients of the plugin's own parameters)."""
        width = int(getattr(self.args, 'task_batch', 0) or 0)
        if width <= 1 or use_second_order:
            return 0
        if not getattr(self.net, 'lockstep_tasks', False):
            # lockstep is OPT-IN per plugin (class attribute `lockstep_tasks = True`): the pass stacks the tasks' samples into
            # one [n*T, ...] batch and hands 5-D stacked weights to MetaConv2dLayer, which is only right for a plugin whose
            # samples never interact (no train-mode BatchNorm) and that reads fast weights through the meta layers only
            return 0
        if not hasattr(self.criterion, 'per_sample'):
            return 0        # a user criterion without per-sample rows: the sequential loop calls it once per task
        if self.args.attenuate and self._routing(frame_shape)[1]:
            return 0
        return width

    def _lockstep_body(self, frames, ids, *, num_steps, msl, training_phase, do_evaluation, importance):
        """What _task_body computes for every task in `ids`, with the T tasks advancing together: activations are
        [n*T, C, H, W] in sample-major order (sample j*T + t belongs to task t), fast weights are stacked [T, *shape], every
        layer is one launch for the whole group (hip_ops.conv_bias_act_tasks), the rule updates the stacked tensors, and the
        criterion is evaluated per sample.  First order only.  Tasks never mix: each task's numbers are those of the
        sequential loop (same kernels per task where the savfi kernels run; MIOpen's grouped solvers elsewhere).

        One difference, invisible in any output: tensors the plugin never reads from the fast dict (SepConv's Subnet /
        Upsample copies) are not carried through step 0.  The reference updates those copies once and never uses them
        (SURVEY.md fact 6); a shared-weight pass cannot give their per-task gradients."""
        T = len(ids)
        routed, _ = self._routing(frames[0].shape[1:])
        theta = self.get_inner_loop_parameter_dict(self.net.named_parameters())
        sel = torch.as_tensor(ids, device=self.device)
        consecutive = list(ids) == list(range(ids[0], ids[0] + T))
        pick = (lambda i: frames[i][ids[0]:ids[0] + T]) if consecutive else (lambda i: frames[i].index_select(0, sel))
        W = {k: theta[k].unsqueeze(0).expand(T, *theta[k].shape).contiguous() for k in routed}
        self.inner_loop_optimizer.initialize_state()
        a, b = self.support_idxs
        sup = [torch.cat([pick(a[i]), pick(b[i])], 0) for i in range(3)]           # frame0 | target | frame1, [2T,3,H,W]
        tgt = [pick(i) for i in self.target_idxs]

        overlap = self.device.type == 'cuda' and bool(getattr(self.args, 'wgrad_overlap', 0))

        def support_loss(weights, num_step):
            model_utils.set_own_params_const(True)      # first-order support pass: the plugin's own parameters are constants
            hip_ops.set_weight_gradient_overlap(overlap)    # weight gradients beside the data-gradient chain; joined below
            try:
                out = self.net.forward(sup[0], sup[2], params=weights, backup_running_statistics=(num_step == 0), num_step=num_step)
            finally:
                model_utils.set_own_params_const(False)
                hip_ops.set_weight_gradient_overlap(False)
            out = out[0] if isinstance(out, tuple) else out
            return self.criterion.per_sample(out, sup[1])['total'].sum()

        def target_pass(weights, num_step):
            out = self.net.forward(tgt[0], tgt[2], params=weights, backup_running_statistics=False, num_step=num_step)
            out = out[0] if isinstance(out, tuple) else out
            return self.criterion.per_sample(out, tgt[1]), out

        if self.args.attenuate:        # L2F per task: embedding [T, L] -> gamma [T, L] -> w_i <- gamma_i * w_i
            keys = list(W)
            loss = support_loss(W, 0)
            grads = torch.autograd.grad(loss, [W[k] for k in keys])
            hip_ops.join_weight_gradients()
            emb = hip_ops.mt_mean([g[t] for g in grads for t in range(T)]).view(len(keys), T).t()
            gamma = 1 - self.gamma_mult * self.attenuator(emb)
            gamma.clamp_(0, 1)
            W = {k: gamma[:, i].reshape((T,) + (1,) * (W[k].dim() - 1)) * W[k] for i, k in enumerate(keys)}

        task_terms, logs, preds = [], [[] for _ in range(T)], None
        for num_step in range(num_steps):
            loss = support_loss(W, num_step)
            keys = list(W)
            grads = torch.autograd.grad(loss, [W[k] for k in keys], allow_unused=True)
            hip_ops.join_weight_gradients()
            W = self.inner_loop_optimizer.update_params(names_weights_dict=W, names_grads_wrt_params_dict=dict(zip(keys, grads)),
                                                        num_step=num_step)
            if msl:
                parts, preds = target_pass(W, num_step)
                task_terms.append(importance[num_step] * parts['total'])
                for k, v in parts.items():
                    for t in range(T):
                        logs[t].append((k, v[t]))
        if not training_phase:
            with torch.no_grad():
                parts, preds = target_pass(W, num_steps)
        elif not msl:
            parts, preds = target_pass(W, num_steps)
        if not training_phase or not msl:
            task_terms.append(parts['total'])
            for k, v in parts.items():
                for t in range(T):
                    logs[t].append((k, v[t]))
        per_task = torch.stack(task_terms, 0).sum(0)                            # [T]
        preds = preds.detach()
        results = []
        for t, task_id in enumerate(ids):
            res = {'pred': self._to_unit_range(preds[t]).unsqueeze(0), 'logs': logs[t], 'loss': per_task[t]}
            if do_evaluation:
                out01 = self._to_unit_range(preds[t])
                tgt01 = self._to_unit_range(frames[self.target_idxs[1]][task_id].detach())
                q_o, q_t = utils.quantize(out01, 1.), utils.quantize(tgt01, 1.)
                res['mse'] = (q_o - q_t).div(255).pow(2).mean()
                res['ssim'] = utils.ssim(q_o.unsqueeze(0), q_t.unsqueeze(0), val_range=255)
            results.append(res)
        if not training_phase:
            self.net.restore_backup_stats()
        return results

    def _run_tasks(self, local, body, flatten=None, streams=None):
        """Results of body(task) for the local tasks, in order.  With --task_streams N > 1 on a GPU the tasks are spread
        over N Python threads, each on its own HIP stream: tasks are independent, and the many small kernels of the deep
        layers (a 12x16 map occupies a fraction of the 256 CUs) then overlap with another task's instead of queueing
        behind each other.  Autograd, the caching allocator and MIOpen handles are per-thread / per-stream safe; the
        per-task rule state and the OWN_PARAMS_CONST flag are thread-local."""
        n = min(self._task_streams(graphed=False) if streams is None else streams, len(local))
        if n <= 1 or self.device.type != 'cuda':
            return [body(t) for t in local]
        dev_index = torch.cuda.current_device()
        if self._task_stream_pool is None or len(self._task_stream_pool) < n:
            self._task_stream_pool = [torch.cuda.Stream(device=dev_index) for _ in range(n)]
        streams = self._task_stream_pool[:n]
        cur = torch.cuda.current_stream(dev_index)
        results, errors = {}, []
        flags = (hip_ops.double_backward(), model_utils.fuse_conv_act())

        def worker(i):
            try:
                torch.cuda.set_device(dev_index)                # new threads start on device 0
                hip_ops.set_double_backward(flags[0])           # per-thread switches: inherit the caller's
                model_utils.set_fuse_conv_act(flags[1])
                with torch.cuda.stream(streams[i]):
                    for t in local[i::n]:
                        results[t] = body(t)
            except BaseException as e:                          # re-raised on the calling thread
                errors.append(e)

        for s in streams:
            s.wait_stream(cur)
        threads = [threading.Thread(target=worker, args=(i,), name="savfi-task-%d" % i) for i in range(n)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        for s in streams:
            cur.wait_stream(s)
        if errors:
            raise errors[0]
        for packed in results.values():                         # produced on a side stream, consumed on the caller's
            for res in ([packed] if flatten is None else flatten(packed)):
                for v in [res['loss'], res['pred'], res.get('mse'), res.get('ssim')] + [v for _, v in res['logs']]:
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(cur)
        return [results[t] for t in local]

    def forward(self, data_batch, epoch, use_second_order, use_multi_step_loss_optimization, num_steps,
                training_phase, do_evaluation=False):
        """Outer-loop forward over the (local shard of the) meta-batch  (reference :346-472).
        Returns (losses, per_task_target_preds, metrics)."""
        frames = data_batch
        num_tasks = len(frames[0])
        self._manual_grads = None
        if (graph_inner_loop.supported(self, use_second_order)
                and not (self.args.attenuate and self._routing(frames[0].shape[1:])[1])      # graphed L2F: every tensor routed
                and self._wants_graphs(frames, training_phase)):
            return self._forward_graphed(frames, epoch, use_multi_step_loss_optimization, num_steps, training_phase,
                                         do_evaluation)
        self._set_pass_flags(use_second_order)
        local = self._local_tasks(num_tasks, training_phase)
        msl = bool(use_multi_step_loss_optimization and training_phase
                   and epoch < self.args.multi_step_loss_num_epochs)

        total_losses = []
        deferred = _DeferredMeters()
        eval_mse, eval_ssim = [], []
        preds = [[] for _ in range(num_tasks)]
        self.net.zero_grad()
        importance = self.get_per_step_loss_importance_vector()

        body = functools.partial(self._task_body, frames, num_steps=num_steps, use_second_order=use_second_order, msl=msl,
                                 training_phase=training_phase, do_evaluation=do_evaluation, importance=importance)
        width = self._lockstep_width(use_second_order, frames[0].shape[1:]) if len(local) > 1 else 0
        if width:
            groups = [local[lo:lo + width] for lo in range(0, len(local), width)]

            def group_body(j):
                group = groups[j]
                if len(group) == 1:
                    return {'group': [body(group[0])]}
                return {'group': self._lockstep_body(frames, group, num_steps=num_steps, msl=msl, training_phase=training_phase,
                                                     do_evaluation=do_evaluation, importance=importance)}
            # --task_streams N > 1: the lockstep groups themselves run concurrently, one thread + HIP stream each
            packed = self._run_tasks(list(range(len(groups))), group_body, flatten=lambda r: r['group'])
            results = [res for p_ in packed for res in p_['group']]
        else:
            results = self._run_tasks(local, body)
        for task_id, res in zip(local, results):
            total_losses.append(res['loss'])
            preds[task_id] = res['pred']
            for k, v in res['logs']:
                deferred.add(k, v)
            if do_evaluation:
                eval_mse.append(res['mse'])
                eval_ssim.append(res['ssim'])

        # mean over the GLOBAL meta-batch: local sum / B (the all-reduce of grads completes the mean)
        if total_losses:
            local_sum = torch.sum(torch.stack(total_losses))
        else:
            local_sum = torch.zeros((), device=self.device)
        losses = {'loss': local_sum / num_tasks}

        metrics = {'psnr': utils.AverageMeter(), 'ssim': utils.AverageMeter()}
        self._logging(losses, metrics, deferred, eval_mse, eval_ssim, importance, training_phase)
        return losses, preds, metrics

    def _wants_graphs(self, frames, training_phase):
        """--graph_inner_loop 1: always (where supported); -1 (default): when this rank would otherwise adapt its tasks ONE AT A
        TIME -- a single local task, or a plugin / configuration outside the lockstep path.  A per-task pass is launch-bound
        (CAIN 64x64: 11 steps/s eager, 31 from graphs); a lockstep pass over several tasks is GPU-bound in eager mode already
        and stays there, where every kernel can be timed in place."""
        mode = int(getattr(self.args, 'graph_inner_loop', 0) or 0)
        if mode > 0:
            return True
        local = self._local_tasks(len(frames[0]), training_phase)
        return len(local) <= 1 or self._lockstep_width(False, frames[0].shape[1:]) <= 1

    def _task_streams(self, graphed):
        """--task_streams N: N tasks (or lockstep groups) in flight, one thread + HIP stream each.  -1 (default): one stream for
        the eager loops -- a lockstep pass fills the GPU by itself -- and up to four for hipGraph replays of single tasks, whose
        deep layers occupy a fraction of the 256 CUs (VoxelFlow 256x256, 8 tasks: 137 -> 160 steps/s)."""
        v = int(getattr(self.args, 'task_streams', 1) or 1)
        if v < 0:
            return 4 if graphed else 1
        return max(1, v)

    def _set_pass_flags(self, use_second_order):
        """Per-pass switches of the op layer (thread-local there: task threads inherit them through _run_tasks)."""
        hip_ops.set_double_backward(bool(use_second_order))
        self._first_order = not use_second_order
        # fused conv epilogues: first-order only (their backward is not differentiable)
        model_utils.set_fuse_conv_act(bool(getattr(self.args, 'fuse_conv_act', 0)) and not use_second_order)

    def _local_tasks(self, num_tasks, training_phase):
        """Training meta-batches are sharded over the ranks (task t -> rank t mod G).  Validation / test sweeps are NOT:
        the data provider hands every rank the full batch (data.py) and ExperimentBuilder consumes every prediction, so each
        rank evaluates all tasks and nothing is reduced -- replicas hold identical weights, hence identical metrics."""
        if not training_phase:
            return list(range(num_tasks))
        return self.task_parallel.local_tasks(num_tasks)

    def _logging(self, losses, metrics, deferred, eval_mse, eval_ssim, importance, training_phase=True):
        """Everything that is logged needs the device to have finished the forward passes.  During training that host sync
        is postponed until the outer backward and the optimizer step are queued (run_train_iter): waiting here would
        drain the queue and the backward would start with the host a whole launch queue behind the GPU."""
        if self._defer_logging:
            importance = importance.detach().clone()      # read later: not the live (learnable) vector an outer step may move
        finish = functools.partial(self._finish_logging, losses, metrics, deferred, eval_mse, eval_ssim, importance,
                                   training_phase)
        if self._defer_logging:
            self._pending_logging = finish
        else:
            finish()

    def _finish_logging(self, losses, metrics, deferred, eval_mse, eval_ssim, importance, training_phase=True):
        """ONE host sync; fills `losses` and `metrics` in place."""
        meters = deferred.flush()
        if eval_mse:
            mse = torch.stack(eval_mse).cpu().tolist()
            ssims = torch.stack(eval_ssim).cpu()
            for m, s in zip(mse, ssims):
                metrics['psnr'].update(-10 * np.log10(m + 1e-8).item())
                metrics['ssim'].update(s)
        if self.task_parallel.active and training_phase:
            self._reduce_logging(losses, meters, metrics)
        for key, meter in meters.items():
            losses[key] = meter.avg
        weights = importance.detach().cpu().numpy()
        for idx in range(len(weights)):
            losses['loss_importance_vector_{}'.format(idx)] = np.asarray(weights[idx])
        # the device has finished everything these numbers depend on: a bounded wait of the wave-specialised SepConv kernels that
        # gave up (csrc/sepconv_ws.hip) left its count in a mapped host word -- wrong numbers must not be logged as a loss
        _hip.ws_check("meta-iteration at epoch %d" % self.current_epoch)

    def _forward_graphed(self, frames, epoch, use_multi_step_loss_optimization, num_steps, training_phase,
                         do_evaluation):
        """Same contract as forward(), executed by hipGraph replays (graph_inner_loop.py): first-order only, the
        outer gradients are assembled by hand and installed by meta_update()."""
        num_tasks = len(frames[0])
        tp = self.task_parallel
        msl = bool(use_multi_step_loss_optimization and training_phase
                   and epoch < self.args.multi_step_loss_num_epochs)
        self._set_pass_flags(False)
        key = (tuple(frames[0].shape[1:]), num_steps, bool(training_phase), msl)
        local = self._local_tasks(num_tasks, training_phase)
        # --task_batch T: groups of T tasks advance in lockstep through ONE graph set (a shorter last group gets its own);
        # --task_streams N: N graph sets per width (own static buffers and memory pool each), replayed from N threads
        width = max(1, self._lockstep_width(False, frames[0].shape[1:]))
        if self.args.attenuate:
            assert not self._routing(frames[0].shape[1:])[1]
            width = 1          # L2F graph sets hold one task (per-task embedding / gamma between the graphs)
        groups = [local[i:i + width] for i in range(0, len(local), width)]
        n = max(1, min(self._task_streams(graphed=True), len(groups)))

        owner = {tuple(g): j % n for j, g in enumerate(groups)}
        needed = {key + (owner[tuple(g)], len(g)) for g in groups}       # graph sets this call replays: never evicted by it

        def loop_for(i, T):
            k = key + (i, T)
            if k not in self._graphs:
                import gc
                # a graph set pins its static buffers and memory pool: keep the most recently used ones only (evaluation over
                # clips of many different sizes would otherwise capture, and keep, one set per size) -- but never one this
                # call still needs (more task streams than MAX_GRAPH_SETS / 2 would otherwise evict their own sets)
                cap = max(self.MAX_GRAPH_SETS, len(needed))
                for old in [q for q in self._graphs if q not in needed]:
                    if len(self._graphs) < cap:
                        break
                    self._graphs.pop(old)
                gc.collect()
                self._graphs[k] = graph_inner_loop.GraphedInnerLoop(self, frames[0].shape[1:], num_steps,
                                                                    bool(training_phase), msl, tasks=T)
            else:
                self._graphs[k] = self._graphs.pop(k)          # most recently used last
            return self._graphs[k]
        # capture / look up on THIS thread, before any worker starts: the workers only read the resolved table
        loops = {tuple(g): loop_for(owner[tuple(g)], len(g)) for g in groups}
        importance = self.get_per_step_loss_importance_vector()
        accums = [graph_inner_loop.OuterGradAccumulator(self, self.get_inner_loop_parameter_dict(self.net.named_parameters()))
                  if training_phase else None for _ in range(n)]

        def body(group):
            i = owner[tuple(group)]
            task_losses, preds_g, logs_g = loops[tuple(group)].run_tasks(frames, list(group), importance, accums[i])
            out = []
            for t, task_id in enumerate(group):
                pred = preds_g[t]
                res = {'loss': task_losses[t], 'pred': self._to_unit_range(pred).unsqueeze(0),
                       'logs': [(k, v) for parts in logs_g[t] for k, v in parts.items()]}
                if do_evaluation:
                    out01 = self._to_unit_range(pred)
                    tgt01 = self._to_unit_range(frames[self.target_idxs[1]][task_id].detach())
                    q_o, q_t = utils.quantize(out01, 1.), utils.quantize(tgt01, 1.)
                    res['mse'] = (q_o - q_t).div(255).pow(2).mean()
                    res['ssim'] = utils.ssim(q_o.unsqueeze(0), q_t.unsqueeze(0), val_range=255)
                out.append(res)
            return out

        if n == 1:
            grouped = [body(g) for g in groups]
        else:
            index = {j: g for j, g in enumerate(groups)}
            by_index = self._run_tasks(list(index), lambda j: {'group': body(index[j])}, flatten=lambda r: r['group'], streams=n)
            grouped = [r['group'] for r in by_index]
            cur = torch.cuda.current_stream()
            for a in accums:                     # summed on the worker streams, merged / installed on this one
                for t in (a.tensors() if a is not None else []):
                    t.record_stream(cur)
        results = [res for group in grouped for res in group]
        accum = accums[0]
        for other in accums[1:]:
            if other is not None:
                accum.merge(other)
        deferred = _DeferredMeters()
        eval_mse, eval_ssim, total_losses = [], [], []
        preds = [[] for _ in range(num_tasks)]
        for task_id, res in zip(local, results):
            total_losses.append(res['loss'])
            preds[task_id] = res['pred']
            for k, v in res['logs']:
                deferred.add(k, v)
            if do_evaluation:
                eval_mse.append(res['mse'])
                eval_ssim.append(res['ssim'])
        local_sum = torch.sum(torch.stack(total_losses)) if total_losses else torch.zeros((), device=self.device)
        losses = {'loss': (local_sum / num_tasks).detach()}
        if training_phase:
            accum.num_tasks = num_tasks
            self._manual_grads = accum
        metrics = {'psnr': utils.AverageMeter(), 'ssim': utils.AverageMeter()}
        self._logging(losses, metrics, deferred, eval_mse, eval_ssim, importance, training_phase)
        return losses, preds, metrics

    def _reduce_logging(self, losses, meters, metrics):
        """Average logging scalars over ranks (second, tiny all-reduce; the loss used for backward stays local)."""
        # fixed key set: a rank whose shard is empty (meta-batch smaller than the world) has no meters of its own
        if hasattr(self.criterion, 'loss_keys'):
            keys = sorted(set(self.criterion.loss_keys()) | set(meters))
        else:                                   # a user criterion: agree on the union of the ranks' keys first
            keys = self.task_parallel.union_of_keys(sorted(meters))
        for k in keys:
            meters.setdefault(k, utils.AverageMeter())
        vec = [float(losses['loss'].detach())]
        for k in keys:
            vec += [float(meters[k].sum), float(meters[k].count)]
        for k in ('psnr', 'ssim'):
            vec += [float(metrics[k].sum), float(metrics[k].count)]
        red = self.task_parallel.allreduce_scalars(torch.tensor(vec, dtype=torch.float64)).cpu().tolist()
        losses['loss_global'] = red[0]
        i = 1
        for k in keys:
            meters[k].sum, meters[k].count = red[i], red[i + 1]
            meters[k].avg = red[i] / max(red[i + 1], 1)
            i += 2
        for k in ('psnr', 'ssim'):
            metrics[k].sum, metrics[k].count = red[i], red[i + 1]
            metrics[k].avg = red[i] / max(red[i + 1], 1)
            i += 2

    # -----------------------------------------------------------------------------------------
    # outer loop entry points (called by ExperimentBuilder)
    # -----------------------------------------------------------------------------------------
    def train_forward_prop(self, data_batch, epoch, do_evaluation=False):
        return self.forward(data_batch=data_batch, epoch=epoch,
                            use_second_order=self.args.second_order and epoch > self.args.first_order_to_second_order_epoch,
                            use_multi_step_loss_optimization=self.args.use_multi_step_loss_optimization,
                            num_steps=self.args.number_of_training_steps_per_iter, training_phase=True,
                            do_evaluation=do_evaluation)

    def evaluation_forward_prop(self, data_batch, epoch):
        return self.forward(data_batch=data_batch, epoch=epoch, use_second_order=False,
                            use_multi_step_loss_optimization=True,
                            num_steps=self.args.number_of_evaluation_steps_per_iter, training_phase=False,
                            do_evaluation=True)

    def _ws_kernels_in_use(self):
        """Does this system's network run the wave-specialised SepConv kernels (the only kernels with bounded in-kernel waits)?"""
        return self.args.model == 'sepconv'

    def meta_update(self, loss):
        """zero_grad -> backward -> (all-reduce of outer grads) -> optimizer step  (reference :551-574)."""
        self.optimizer.zero_grad()
        params = list(self.trainable_parameters())
        # task parallelism: .grad = zeroed views of the flat all-reduce bucket, filled in place by the backward pass
        views = self.task_parallel.prepare_gradients(params)
        if self._manual_grads is not None:        # graphed forward: first-order outer gradients assembled by hand
            reached = self._manual_grads.install(self._manual_grads.num_tasks, into=views)
            if views is not None:
                self.task_parallel.mark_touched(reached)
            self._manual_grads = None
        elif loss.requires_grad:
            loss.backward()
        self.task_parallel.allreduce_gradients(params)
        if self.device.type == 'cuda' and self._ws_kernels_in_use() and _hip.ws_armed():
            # Fail closed: a bounded wait of the wave-specialised SepConv kernels that gave up (csrc/sepconv_ws.hip) has produced wrong
            # gradients.  The stream is drained HERE, before theta moves -- one host wait per meta-iteration, ~0.3 % of a C2 iteration
            # (profiles/r06_ws_check_cost.txt) -- and the exception leaves theta, the optimizer state and the scheduler untouched.
            torch.cuda.current_stream(self.device).synchronize()
            _hip.ws_check("this meta-iteration at epoch %d: its outer step was NOT applied" % self.current_epoch)
        self.optimizer.step()
        if self.device.type == 'cuda':      # the filters every conv layer keeps of its own weight: one launch for all of them
            if self._filter_modules is None:
                self._filter_modules = [m for m in self.net.modules() if isinstance(m, model_utils.MetaConv2dLayer)]
            hip_ops.refresh_module_filters(self._filter_modules)

    def run_train_iter(self, data_batch, epoch, do_evaluation=False):
        _hip.ws_check("an earlier iteration")       # one host word: what a caller that never reads its losses would otherwise miss
        epoch = int(epoch)
        self.current_epoch = epoch
        if not self.training:
            self.train()
        data_batch = [frame.to(device=self.device, non_blocking=True) for frame in data_batch]
        self._defer_logging = True
        try:
            losses, preds, metrics = self.train_forward_prop(data_batch=data_batch, epoch=epoch,
                                                             do_evaluation=do_evaluation)
            self.meta_update(loss=losses['loss'])
            self.optimizer.zero_grad()
            self.zero_grad()
        finally:
            self._defer_logging = False
            finish, self._pending_logging = self._pending_logging, None
        if finish is not None:
            if self.task_parallel.active or not getattr(self.args, 'lazy_logging', 1):
                finish()          # the logging all-reduce is a collective: every rank issues it here, in program order
            else:
                state = {'finish': finish}

                def ensure():
                    f = state.pop('finish', None)
                    if f is not None:
                        f()
                losses, metrics = _LazyLog(losses, ensure), _LazyLog(metrics, ensure)
        return losses, preds, metrics

    def run_validation_iter(self, data_batch):
        _hip.ws_check("an earlier iteration")
        data_batch = [frame.to(device=self.device, non_blocking=True) for frame in data_batch]
        return self.evaluation_forward_prop(data_batch=data_batch, epoch=self.current_epoch)

    def run_test_iter(self, data_batch):
        """Adapt on a 4-frame clip ((0,2)->1, (1,3)->2), then interpolate between frames 1 and 2
        (reference :630-697).  Returns a list of [3,H,W] predictions."""
        _hip.ws_check("an earlier iteration")
        if self.training:
            self.eval()
        frames = [frame.to(device=self.device, non_blocking=True) for frame in data_batch]
        saved = self.support_idxs
        self.support_idxs = [[0, 1, 2], [1, 2, 3]]
        preds = [[] for _ in range(len(frames[0]))]
        self.net.zero_grad()
        # test-time adaptation never back-propagates through the inner loop: first-order passes whatever --second_order says
        # (the reference passes create_graph=args.second_order here, :664, and then discards the graph)
        self._set_pass_flags(False)
        try:
            for task_id in range(len(frames[0])):
                steps = self.args.number_of_evaluation_steps_per_iter
                weights = self._adapt(frames, task_id, steps, False)
                with torch.no_grad():
                    out = self.net.forward(frames[1][task_id].unsqueeze(0), frames[2][task_id].unsqueeze(0),
                                           params=weights, backup_running_statistics=False,
                                           num_step=max(steps - 1, 0))
                if isinstance(out, tuple):     # superslomo (:686-688)
                    out = self.revNormalize(out[0].squeeze(0)).unsqueeze(0)
                preds[task_id] = out.squeeze(0).detach()
                self.net.restore_backup_stats()
        finally:
            self.support_idxs = saved
        return preds
