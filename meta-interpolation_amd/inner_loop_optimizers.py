"""Inner-loop learning rules: LSLR (per-layer per-step lr) and Meta-SGD (per-element lr).

Same surface as the reference (inner_loop_optimizers.py:57-244, :248-425):

    Rule(device, optimizer in {'SGD','Adam','Adamax'}, init_learning_rate[, total_num_inner_loop_steps,
         use_learnable_learning_rates])
    .initialize(names_weights_dict)   -> builds .names_learning_rates_dict (nn.ParameterDict keyed by
                                         name.replace('.', '-'))
    .initialize_state()               -> fresh per-task moments
    .update_params(names_weights_dict, names_grads_wrt_params_dict, num_step) -> new dict

but one update is ONE fused multi-tensor launch per <=48 tensors (savfi_mt_update_f32) instead of
2..12 launches per tensor.  The arithmetic is the reference's *as implemented* (SURVEY.md section 0,
fact 5c): LSLR-Adamax keeps a first moment and divides by |g|+eps, Meta-SGD-Adamax is
lr * (1-b1)/(1-b1^t) * g/(|g|+eps); Adam uses betas (0.9, 0.99), eps 1e-8, per-key step counts.

Behaviour on the reference's crashing configurations (documented divergence):
  * a None gradient skips the parameter for every rule (the reference's Meta-SGD+SGD raises TypeError);
  * d w'/d lr is captured by the forward kernel, so learnable learning rates work with Adam/Adamax
    for any number of steps (the reference fails in the outer backward for >= 2 steps).
"""
import math

import threading

import torch
import torch.nn as nn

from . import _hip, hip_ops


def _lr_key(name):
    return name.replace(".", "-")


class _FusedInnerRule(nn.Module):
    beta1, beta2, eps, weight_decay = 0.9, 0.99, 1e-8, 0

    def __init__(self, optimizer):
        super().__init__()
        self.optimizer = optimizer
        object.__setattr__(self, '_tls', threading.local())      # per-task moments live per THREAD (concurrent tasks)
        self.names_learning_rates_dict = nn.ParameterDict()

    @property
    def state(self):
        st = getattr(self._tls, 'state', None)
        if st is None:
            st = self._tls.state = {}
        return st

    @state.setter
    def state(self, value):
        self._tls.state = value

    # -- surface ---------------------------------------------------------------------------
    def initialize_state(self):
        """Forget the moments: called once per task (meta_learning_system.py:377)."""
        self.state = {}

    def reset(self):
        pass

    def update_params(self, names_weights_dict, names_grads_wrt_params_dict, num_step, tau=0.1):
        if self.optimizer not in ('SGD', 'Adam', 'Adamax'):
            raise NotImplementedError('This type of optimizer update operation is not yet implemented')
        keys = [k for k, g in names_grads_wrt_params_dict.items() if g is not None]
        if not keys:
            return dict()
        ws = [names_weights_dict[k] for k in keys]
        gs = [names_grads_wrt_params_dict[k] for k in keys]
        lrs = [self._lr(k, num_step) for k in keys]
        if any(g.requires_grad for g in gs):
            # second-order MAML: the update must stay differentiable w.r.t. g -> composed device ops
            new = self._update_composed(keys, ws, gs, lrs)
        else:
            new = self._update_fused(keys, ws, gs, lrs)
        return dict(zip(keys, new))

    # -- fused first-order path --------------------------------------------------------------
    def _moments(self, keys, ws, names):
        out = {n: [] for n in names}
        bc1, sbc2 = [], []
        for k, w in zip(keys, ws):
            st = self.state.setdefault(k, {'step': 0})
            st['step'] += 1
            for n in names:
                if n not in st:
                    st[n] = torch.zeros_like(w, memory_format=torch.contiguous_format)
                out[n].append(st[n])
            bc1.append(1 - self.beta1 ** st['step'])
            sbc2.append(math.sqrt(1 - self.beta2 ** st['step']))
        return out, bc1, sbc2

    def _update_fused(self, keys, ws, gs, lrs):
        gs = [g if g.is_contiguous() else g.contiguous() for g in gs]
        hyper = dict(beta1=self.beta1, beta2=self.beta2, eps=self.eps)
        if self.optimizer == 'SGD':
            return hip_ops.mt_update(_hip.RULE_SGD, self.lr_mode, ws, gs, lrs, **hyper)
        if self.optimizer == 'Adam':
            mom, bc1, sbc2 = self._moments(keys, ws, ('exp_avg', 'exp_avg_sq'))
            return hip_ops.mt_update(_hip.RULE_ADAM, self.lr_mode, ws, gs, lrs, m=mom['exp_avg'],
                                     s=mom['exp_avg_sq'], bc1=bc1, sqrt_bc2=sbc2, **hyper)
        if self.keeps_adamax_moment:
            mom, bc1, _ = self._moments(keys, ws, ('exp_avg',))
            return hip_ops.mt_update(_hip.RULE_ADAMAX_LSLR, self.lr_mode, ws, gs, lrs, m=mom['exp_avg'],
                                     bc1=bc1, **hyper)
        _, bc1, _ = self._moments(keys, ws, ())
        return hip_ops.mt_update(_hip.RULE_ADAMAX_MSGD, self.lr_mode, ws, gs, lrs, bc1=bc1, **hyper)

    # -- composed path (only when create_graph=True made the grads differentiable) -------------
    def _update_composed(self, keys, ws, gs, lrs):
        b1, b2, eps = self.beta1, self.beta2, self.eps
        out = []
        for k, w, g, lr in zip(keys, ws, gs, lrs):
            if self.optimizer == 'SGD':
                out.append(w - lr * g)
                continue
            st = self.state.setdefault(k, {'step': 0})
            st['step'] += 1
            bc1 = 1 - b1 ** st['step']
            if self.optimizer == 'Adam':
                m = st.get('exp_avg', 0) * b1 + (1 - b1) * g
                s = st.get('exp_avg_sq', 0) * b2 + (1 - b2) * g * g
                st['exp_avg'], st['exp_avg_sq'] = m, s
                denom = s.sqrt() / math.sqrt(1 - b2 ** st['step']) + eps
                out.append(w - (lr / bc1) * m / denom)
            elif self.keeps_adamax_moment:
                m = st.get('exp_avg', 0) * b1 + (1 - b1) * g
                st['exp_avg'] = m
                out.append(w - (lr / bc1) * m / (g.abs() + eps))
            else:
                out.append(w - (lr / bc1) * ((1 - b1) * g) / (g.abs() + eps))
        return out


class LSLRGradientDescentLearningRule(_FusedInnerRule):
    """Per-layer, per-step learning rates (MAML++ LSLR); reference :57-244."""
    lr_mode = _hip.LR_SCALAR
    keeps_adamax_moment = True

    def __init__(self, device, optimizer, total_num_inner_loop_steps, use_learnable_learning_rates,
                 init_learning_rate=1e-3):
        super().__init__(optimizer)
        self.device = device
        self.init_learning_rate = float(init_learning_rate)
        self.total_num_inner_loop_steps = total_num_inner_loop_steps
        self.use_learnable_learning_rates = use_learnable_learning_rates

    def initialize(self, names_weights_dict):
        self.names_learning_rates_dict = nn.ParameterDict()
        for key in names_weights_dict.keys():
            table = torch.ones(self.total_num_inner_loop_steps + 1, device=self.device) * self.init_learning_rate
            self.names_learning_rates_dict[_lr_key(key)] = nn.Parameter(
                table, requires_grad=self.use_learnable_learning_rates)

    def _lr(self, key, num_step):
        return self.names_learning_rates_dict[_lr_key(key)][num_step]


class MetaSGDLearningRule(_FusedInnerRule):
    """One learnable learning rate per parameter element (Meta-SGD); reference :248-425."""
    lr_mode = _hip.LR_ELEMENT
    keeps_adamax_moment = False

    def __init__(self, device, optimizer, init_learning_rate=1e-3):
        super().__init__(optimizer)
        assert init_learning_rate > 0., 'learning_rate should be positive.'
        self.device = device
        self.init_learning_rate = float(init_learning_rate)

    def initialize(self, names_weights_dict):
        self.names_learning_rates_dict = nn.ParameterDict()
        for key, param in names_weights_dict.items():
            self.names_learning_rates_dict[_lr_key(key)] = nn.Parameter(
                torch.ones_like(param, device=self.device) * self.init_learning_rate, requires_grad=True)

    def reset(self):
        with torch.no_grad():
            for p in self.names_learning_rates_dict.values():
                p.fill_(self.init_learning_rate)

    def _lr(self, key, num_step):
        return self.names_learning_rates_dict[_lr_key(key)]
