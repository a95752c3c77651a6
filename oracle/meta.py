"""oracle/meta.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's per-task MAML inner loop and outer loss
(SceneAdaptiveInterpolation.forward, meta_learning_system.py:346-472, with :186-210, :213-228,
:231-272, :275-321, :475-509), written as one flat function over explicit tensors so that it can be
compared step by step with both the imported reference (oracle/gen_golden.py -> tests/golden) and
the HIP product path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import models, rules

SUPPORT = [[0, 2, 4], [2, 4, 6]]   # meta_learning_system.py:43
TARGET = [2, 3, 4]                 # :46


def criterion(kind):
    """Loss wrapper for '1*L1' / '1*MSE' (loss.py:287-290, :325-350; hr.clone() is value-neutral)."""
    fn = F.l1_loss if kind == 'L1' else F.mse_loss
    return lambda out, tgt: 1.0 * fn(out, tgt)


def importance_vector(S, epoch, msl_epochs):
    """get_per_step_loss_importance_vector, :186-210."""
    if S == 0:
        return torch.ones(1)
    w = np.ones(shape=(S)) * (1.0 / S)
    decay_rate = 1.0 / S / msl_epochs
    min_value = 0.03 / S
    for i in range(len(w) - 1):
        w[i] = np.maximum(w[i] - (epoch * decay_rate), min_value)
    w[-1] = np.minimum(w[-1] + (epoch * (S - 1) * decay_rate), 1.0 - ((S - 1) * min_value))
    return torch.Tensor(w)


def inner_param_names(params, enable_bn=False):
    """get_inner_loop_parameter_dict, :213-228 (params: iterable of (name, tensor))."""
    return [n for n, p in params if p.requires_grad and (enable_bn or 'norm_layer' not in n)]


def fingerprint(t):
    t = t.detach().double().reshape(-1)
    return [t.sum().item(), t.abs().sum().item()] + t[:4].tolist()


def run_iteration(model, base, frames, *, rule='lslr', optimizer='SGD', lrs, num_steps, loss='L1',
                  training=True, second_order=False, msl=False, epoch=0, msl_epochs=1,
                  attenuator=None, gamma_mult=None, forward_kwargs=None, record=None):
    """One meta-iteration over all tasks of `frames` (list of 7 tensors [B,3,H,W]).

    base: {name: tensor} parameters (leaf, requires_grad where trainable) + buffers of the backbone.
    lrs : rules.init_lrs(...) learning-rate tensors.
    Returns dict(loss=mean task loss (graph attached when training), preds=[...], task_losses=[...]).
    `record` (optional dict) receives per-task, per-step fingerprints of grads and fast weights.
    """
    fwd = models.FORWARD[model]
    kw = forward_kwargs or {}
    crit = criterion(loss)
    names = inner_param_names([(n, p) for n, p in base.items() if p.is_floating_point()], False)
    B = frames[0].shape[0]
    imp = importance_vector(num_steps if training else num_steps, epoch, msl_epochs) if training else None
    use_msl = bool(msl and training and epoch < msl_epochs)
    total, preds = [], []
    for t in range(B):
        fast = {n: base[n] for n in names}                      # :370-372 (the Parameters themselves)
        st = rules.RuleState()                                   # :377 initialize_state
        pair = lambda ind, w: crit(fwd(frames[ind[0]][t][None], frames[ind[2]][t][None], base, w, **kw),
                                   frames[ind[1]][t][None])
        if attenuator is not None:                               # L2F, :231-272
            sl = pair(SUPPORT[0], fast) + pair(SUPPORT[1], fast)
            g = torch.autograd.grad(sl, list(fast.values()), create_graph=False, allow_unused=True)
            emb = torch.stack([gi.mean() for gi in g])
            if record is not None:
                record.setdefault('embedding', []).append(emb.detach().clone())
            gamma = 1 - gamma_mult * attenuator(emb)
            gamma = gamma.clamp(0, 1)
            fast = {n: gamma[i] * w for i, (n, w) in enumerate(fast.items())}
        task_losses = []
        pred = None
        for step in range(num_steps):                            # :386-412
            sl = pair(SUPPORT[0], fast) + pair(SUPPORT[1], fast)
            g = torch.autograd.grad(sl, list(fast.values()), create_graph=second_order, allow_unused=True)
            grads = dict(zip(fast.keys(), g))
            fast = rules.update_params(rule, optimizer, fast, grads, lrs, step, st)
            if record is not None:
                record.setdefault('support_loss', []).append(sl.item())
                record.setdefault('n_live', []).append(len(fast))
                record.setdefault('grad_fp', []).append(
                    {k: fingerprint(v) for k, v in grads.items() if v is not None})
                record.setdefault('weight_fp', []).append({k: fingerprint(v) for k, v in fast.items()})
            if use_msl:
                pred = fwd(frames[TARGET[0]][t][None], frames[TARGET[2]][t][None], base, fast, **kw)
                task_losses.append(imp[step] * crit(pred, frames[TARGET[1]][t][None]))
        if not training:                                         # :414-423
            with torch.no_grad():
                pred = fwd(frames[TARGET[0]][t][None], frames[TARGET[2]][t][None], base, fast, **kw)
                task_losses.append(crit(pred, frames[TARGET[1]][t][None]))
        elif not use_msl:                                        # :424-432
            pred = fwd(frames[TARGET[0]][t][None], frames[TARGET[2]][t][None], base, fast, **kw)
            task_losses.append(crit(pred, frames[TARGET[1]][t][None]))
        preds.append(pred.detach())
        total.append(torch.sum(torch.stack(task_losses)))        # :460-461
    loss_val = torch.mean(torch.stack(total))                    # :338
    return dict(loss=loss_val, preds=preds, task_losses=[x.detach() for x in total])


def psnr(pred01, gt01):
    """utils.quantize + calc_psnr, utils.py:171-186 (inputs in [0,1], [3,H,W])."""
    import math
    q = lambda x: x.mul(255).clamp(0, 255).round()
    diff = (q(pred01) - q(gt01)).div(255)
    return -10 * math.log10(diff.pow(2).mean() + 1e-8)


def run_test_iteration(model, base, frames, *, rule='lslr', optimizer='SGD', lrs, num_steps, loss='L1',
                       forward_kwargs=None):
    """run_test_iter, meta_learning_system.py:630-697: per clip of 4 frames, adapt on (0,2)->1 and (1,3)->2
    (:653), then interpolate between frames 1 and 2 with the adapted weights (:680-684)."""
    fwd = models.FORWARD[model]
    kw = forward_kwargs or {}
    crit = criterion(loss)
    names = inner_param_names([(n, p) for n, p in base.items() if p.is_floating_point()], False)
    support = [[0, 1, 2], [1, 2, 3]]
    preds = []
    for t in range(frames[0].shape[0]):
        fast = {n: base[n] for n in names}
        st = rules.RuleState()
        for step in range(num_steps):
            sl = 0
            for ind in support:
                sl = sl + crit(fwd(frames[ind[0]][t][None], frames[ind[2]][t][None], base, fast, **kw),
                               frames[ind[1]][t][None])
            g = torch.autograd.grad(sl, list(fast.values()), create_graph=False, allow_unused=True)
            fast = rules.update_params(rule, optimizer, fast, dict(zip(fast.keys(), g)), lrs, step, st)
        with torch.no_grad():
            out = fwd(frames[1][t][None], frames[2][t][None], base, fast, **kw).squeeze(0)
            if model == 'superslomo':        # revNormalize, :686-688 (Normalize(mean=-m, std=1): x + m)
                out = out + torch.tensor([0.429, 0.431, 0.397]).view(3, 1, 1)
            preds.append(out)
    return preds
