"""hipGraph-captured first-order inner loop.

In eager mode one inner step of SepConv at 256x448 issues ~600 kernel launches from Python (forward,
autograd, fused update); the host needs ~8 ms for them, and for the low-resolution / many-layer
configurations (CAIN: ~8k launches per iteration) the GPU simply waits for the host.  First-order MAML
makes the whole step capturable:

    support step t :  W_t  --[ N=2 support forward, loss, autograd.grad, fused update ]-->  W_{t+1}   (one hipGraph)
    target pass    :  W_S  --[ forward, loss, grads w.r.t. W_S and the non-routed theta ]-->  G_S      (one hipGraph)

and, because d W_{t+1} / d W_t = I when the gradients are constants, the outer gradient needs NO autograd
graph across steps:   dL/d theta_k = sum_s w_s G_s,k   (routed tensors; identity chain W_S -> ... -> W_0 = theta),
dL/d lr_t = < sum_{s>t} w_s G_s , dir_t >   with dir_t = d W_{t+1} / d lr_t (= -g_t for SGD, the kernel's `coef` else).
Every graph replays on static buffers: step graph t reads the tensors graph t-1 wrote.  All graphs share one
memory pool (they never run concurrently).  The numerics are those of the eager path (same kernels, same order).

Tasks in lockstep (``tasks`` = T > 1, --task_batch): the same graphs over T tasks at once -- support batch [2T,3,H,W] in
sample-major order, fast weights / moments / gradients stacked [T, *shape], per-sample losses; the identity chain then sums
the stacked target gradients over the task axis.  T = 1 is the per-task form above, unchanged.

L2F (--attenuate, T = 1, plugins that route every inner-loop tensor): one more graph in front -- the support pass at theta and
its layer-wise mean gradients (the task embedding) -- then, eagerly, the attenuator MLP (two tiny GEMVs, autograd on) and
W_0 = gamma * theta; the hand-assembled outer gradient gains d/d theta_k = gamma_k * sum_s w_s G_s,k and d/d gamma_k =
<sum_s w_s G_s,k, theta_k>, which autograd carries through the attenuator to its parameters and gamma_mult.

Not captured (the caller falls back to the eager loop): --second_order, L2F on a plugin with unrouted tensors or T > 1, CPU tensors.
"""
import contextlib

import torch

from . import _hip, hip_ops, model_utils, utils


# Graphed L2F is OPT-IN (SAVFI_GRAPH_L2F=1).  It matches the eager loop and the reference fixtures from 64x64 to 1280x720
# (tests/test_system_gpu.py, tests/test_fullsize_gpu.py::test_graphed_cain_720p_follows_changing_frames,
# profiles/r03_graphed_l2f_720p.txt) -- the garbage its third replay returned at 720p earlier this round was a stale captured
# mean (ATen's x.mean(2) behind a hipGraph memset node, which only clears once on ROCm 7.2: csrc/submean.hip) -- but config C5 is
# GPU-bound and gains nothing from it (6.93 vs 6.91 steps/s), so the default keeps the eager loop, where kernels can be timed in place.
import os as _os
GRAPH_L2F = False      # tools/graph_vs_eager.py, tools/memset_capture_audit.py set it: L2F from graph replays (DESIGN: the ROCm 7.2 memset-node finding)


def supported(system, use_second_order):
    a = system.args
    if a.attenuate and (system._routing_known_incomplete() or not GRAPH_L2F):
        return False
    return (bool(getattr(a, 'graph_inner_loop', 0)) and system.device.type == 'cuda' and not use_second_order
            and hasattr(system.inner_loop_optimizer, 'lr_mode'))


def _frame(out):
    """Super SloMo's forward returns (frame, extras for its 'Super' loss): the frame is what this path uses."""
    return out[0] if isinstance(out, tuple) else out


class GraphedInnerLoop:
    def __init__(self, system, frame_shape, num_steps, training, msl, tasks=1):
        self.sys = system
        self.T = int(tasks)
        self.net, self.rule, self.crit = system.net, system.inner_loop_optimizer, system.criterion
        self.S, self.training, self.msl = num_steps, training, msl
        self.shape = tuple(frame_shape)            # (3, H, W)
        dev = system.device
        named = system.get_inner_loop_parameter_dict(self.net.named_parameters())
        self.all_keys = list(named.keys())
        self.theta = named
        self._probe_routing()
        C, H, W = self.shape
        T = self.T
        self.sup = [torch.zeros(2 * T, C, H, W, device=dev) for _ in range(3)]      # frame0 | target | frame1, pair batch
        self.tgt = [torch.zeros(T, C, H, W, device=dev) for _ in range(3)]
        stacked = (lambda p: torch.zeros_like(p)) if T == 1 else (lambda p: torch.zeros((T,) + tuple(p.shape), device=dev))
        self._like = stacked
        self.W0 = {k: stacked(self.theta[k]).requires_grad_() for k in self.routed}
        self.learn_lr = any(p.requires_grad for p in self.rule.names_learning_rates_dict.values())
        opt = self.rule.optimizer
        self.rule_id = {('SGD', True): _hip.RULE_SGD, ('SGD', False): _hip.RULE_SGD,
                        ('Adam', True): _hip.RULE_ADAM, ('Adam', False): _hip.RULE_ADAM,
                        ('Adamax', True): _hip.RULE_ADAMAX_LSLR, ('Adamax', False): _hip.RULE_ADAMAX_MSGD}[
            (opt, self.rule.keeps_adamax_moment)]
        self.m = [stacked(self.theta[k]) for k in self.routed] if self.rule_id in (1, 2) else None
        self.s = [stacked(self.theta[k]) for k in self.routed] if self.rule_id == 1 else None
        self.attenuate = bool(getattr(system.args, 'attenuate', False))
        if self.attenuate:
            assert T == 1 and not self.unrouted, "graphed L2F: one task per graph set, every inner-loop tensor routed"
        self.emb_graph, self.emb_out = None, None    # L2F: support pass at theta -> layer-wise mean gradients
        self.step_graphs, self.step_out = [], []     # per step: graph, dict(W_out, g, dir)
        self._theta_to_w0 = None                     # cached pointer tables of the theta -> W_0 copy (T = 1)
        self.target_graphs = {}                      # step index s (params = W_s) -> (graph, outputs)
        self.pool = None
        self._capture()

    # ------------------------------------------------------------------------------------------
    def _probe_routing(self):
        """Which inner-loop tensors does the plugin actually read from the fast dict?  (SepConv 54 of 94,
        VoxelFlow 9 of 23, CAIN 494 of 494: SURVEY.md fact 6.)  One eager forward on cloned tensors."""
        C, H, W = self.shape
        dev = self.sys.device
        fast = {k: v.detach().clone().requires_grad_() for k, v in self.theta.items()}
        x = torch.zeros(1, C, H, W, device=dev)
        out = _frame(self.net.forward(x, x, params=fast, backup_running_statistics=False, num_step=0))
        g = torch.autograd.grad(out.sum(), list(fast.values()), allow_unused=True)
        self.routed = [k for k, gi in zip(self.all_keys, g) if gi is not None]
        self.unrouted = [k for k, gi in zip(self.all_keys, g) if gi is None]

    def _lrs(self, t):
        return [self.rule._lr(k, t) for k in self.routed]

    def _support_step(self, W, t):
        if t == 0:
            # W_0 is not the output of an update: give its layers the one-launch-per-kind filter transform the later steps get from
            # mt_update (the plan is keyed by the list's shapes, which W_0 shares with every W_t); 127 single-layer launches per
            # replay of CAIN's step graph otherwise
            hip_ops.filters_after_update([W[k] for k in self.routed])
        model_utils.set_own_params_const(True)      # first-order support pass: non-routed parameters are constants
        try:
            out = _frame(self.net.forward(self.sup[0], self.sup[2], params=W, backup_running_statistics=(t == 0), num_step=t))
        finally:
            model_utils.set_own_params_const(False)
        if self.T == 1:
            loss = self.crit(out[0:1], self.sup[1][0:1])['total'] + self.crit(out[1:2], self.sup[1][1:2])['total']
        else:
            loss = self.crit.per_sample(out, self.sup[1])['total'].sum()
        g = torch.autograd.grad(loss, [W[k] for k in self.routed])
        ws = [W[k].detach() for k in self.routed]
        lrs = [l.detach() for l in self._lrs(t)]
        bc1 = [1 - self.rule.beta1 ** (t + 1)] * len(ws)
        sbc2 = [(1 - self.rule.beta2 ** (t + 1)) ** 0.5] * len(ws)
        want_coef = self.learn_lr and self.rule_id != _hip.RULE_SGD
        new, coef = hip_ops.mt_update_nograd(self.rule_id, self.rule.lr_mode, ws, list(g), lrs, self.m, self.s, bc1, sbc2,
                                             self.rule.beta1, self.rule.beta2, self.rule.eps, want_coef)
        Wn = {k: v.requires_grad_() for k, v in zip(self.routed, new)}
        return dict(W=Wn, g=list(g), dir=(coef if want_coef else list(g)), loss=loss.detach())

    def _embedding(self, W):
        """L2F task embedding (reference meta_learning_system.py:231-255): support loss at theta, first-order gradients, one
        fused per-tensor mean."""
        model_utils.set_own_params_const(True)
        try:
            out = _frame(self.net.forward(self.sup[0], self.sup[2], params=W, backup_running_statistics=True, num_step=0))
        finally:
            model_utils.set_own_params_const(False)
        loss = self.crit(out[0:1], self.sup[1][0:1])['total'] + self.crit(out[1:2], self.sup[1][1:2])['total']
        g = torch.autograd.grad(loss, [W[k] for k in self.routed])
        return hip_ops.mt_mean(g)

    def _target(self, W, s, with_grad):
        crit = self.crit if self.T == 1 else self.crit.per_sample          # T > 1: every loss part is a [T] vector
        if not with_grad:
            with torch.no_grad():
                pred = _frame(self.net.forward(self.tgt[0], self.tgt[2], params=W, backup_running_statistics=False, num_step=s))
                parts = crit(pred, self.tgt[1])
            return dict(pred=pred, parts={k: v.detach() for k, v in parts.items()})
        pred = _frame(self.net.forward(self.tgt[0], self.tgt[2], params=W, backup_running_statistics=False, num_step=s))
        parts = crit(pred, self.tgt[1])
        own = [self.theta[k] for k in self.unrouted]
        g = torch.autograd.grad(parts['total'].sum(), [W[k] for k in self.routed] + own, allow_unused=True)
        n = len(self.routed)
        return dict(pred=pred.detach(), parts={k: v.detach() for k, v in parts.items()}, g_routed=list(g[:n]),
                    g_own=list(g[n:]))

    def _capture(self):
        need_target = sorted(set(range(1, self.S + 1)) if (self.msl and self.training) else {self.S})
        with_grad = self.training

        def run_all():
            W = self.W0
            outs, tg = [], {}
            if self.attenuate:
                self._embedding(W)
            for t in range(self.S):
                o = self._support_step(W, t)
                outs.append(o)
                W = o['W']
                if (t + 1) in need_target:
                    tg[t + 1] = self._target(W, t + 1, with_grad)
            if 0 in need_target or self.S == 0:
                tg[self.S] = self._target(W, self.S, with_grad)
            return outs, tg

        # warm-up on a side stream (MIOpen find, lazy inits), never on the default stream
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                run_all()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

        self.pool = torch.cuda.graph_pool_handle()
        W = self.W0
        if self.attenuate:
            with self._captured('embedding') as g:
                self.emb_out = self._embedding(W)
            self.emb_graph = g
        for t in range(self.S):
            with self._captured('step%d' % t) as g:
                o = self._support_step(W, t)
            self.step_graphs.append(g)
            self.step_out.append(o)
            W = o['W']
            if (t + 1) in need_target:
                with self._captured('target%d' % (t + 1)) as tgph:
                    to = self._target(W, t + 1, with_grad)
                self.target_graphs[t + 1] = (tgph, to)
        if self.S == 0:
            with self._captured('target0') as tgph:
                to = self._target(W, 0, with_grad)
            self.target_graphs[0] = (tgph, to)

    @contextlib.contextmanager
    def _captured(self, tag):
        """One capture into the shared pool."""
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self.pool):
            yield g

    # ------------------------------------------------------------------------------------------
    def run_task(self, frames, task_id, importance, accum):
        """Adapt on one task (T = 1) and (when training) add its outer-gradient contribution to `accum`.
        Returns (task_loss scalar tensor, pred [1,3,H,W], list of loss-part dicts)."""
        losses, preds, logs = self.run_tasks(frames, [task_id], importance, accum)
        return losses[0], preds[0:1], logs[0]

    def run_tasks(self, frames, ids, importance, accum):
        """Adapt the T tasks `ids` together; returns (task losses [T], preds [T,3,H,W], per task a list of loss-part dicts)."""
        sysm = self.sys
        T = self.T
        assert len(ids) == T, (ids, T)
        a, b = sysm.support_idxs
        tix = sysm.target_idxs
        if list(ids) == list(range(ids[0], ids[0] + T)):
            pick = lambda i: frames[i][ids[0]:ids[0] + T]
        else:
            sel = torch.as_tensor(list(ids), device=frames[0].device)
            pick = lambda i: frames[i].index_select(0, sel)
        for dst, (ia, ib) in zip(self.sup, ((a[0], b[0]), (a[1], b[1]), (a[2], b[2]))):
            dst[0:T].copy_(pick(ia))           # sample-major: sample j * T + t belongs to task t
            dst[T:2 * T].copy_(pick(ib))
        for dst, i in zip(self.tgt, tix):
            dst.copy_(pick(i))
        with torch.no_grad():
            if T == 1:
                if self._theta_to_w0 is None:
                    self._theta_to_w0 = hip_ops.MtCopy([self.W0[k] for k in self.routed], [self.theta[k] for k in self.routed])
                self._theta_to_w0.run()
            else:
                for k in self.routed:
                    self.W0[k].copy_(self.theta[k].unsqueeze(0).expand_as(self.W0[k]))
            for buf in (self.m, self.s):
                if buf is not None:
                    torch._foreach_zero_(buf)
        logs, task_loss, pred = [], None, None
        gamma = None
        if self.attenuate:
            # L2F: embedding from a replay, gamma eagerly (autograd on: the attenuator's parameters and gamma_mult get their
            # outer gradient through it), W_0 = gamma * theta into the static buffers
            self.emb_graph.replay()
            with torch.enable_grad():
                gamma = (1 - sysm.gamma_mult * sysm.attenuator(self.emb_out.detach())).clamp(0, 1)
            with torch.no_grad():
                hip_ops.mt_scale_into(gamma, [self.theta[k] for k in self.routed], [self.W0[k] for k in self.routed])
        # replay: S support steps, target passes where needed
        for t in range(self.S):
            self.step_graphs[t].replay()
            if (t + 1) in self.target_graphs:
                self.target_graphs[t + 1][0].replay()
        if self.S == 0:
            self.target_graphs[0][0].replay()

        # losses / outer gradients from the static outputs
        msl = self.msl and self.training
        weights = {s: (importance[s - 1] if msl else 1.0) for s in self.target_graphs}
        for s, (_, to) in sorted(self.target_graphs.items()):
            w = weights[s]
            term = w * to['parts']['total']
            task_loss = term if task_loss is None else task_loss + term
            # the parts are STATIC output buffers of the captured graph: a later replay of the same graph set (another group of
            # the same call, the next meta-iteration behind a lazy log) would overwrite what the caller logs -- one stacked
            # clone per target graph (a handful of scalars)
            keys = list(to['parts'])
            vals = [to['parts'][k] for k in keys]
            if all(v.shape == vals[0].shape and v.dtype == vals[0].dtype for v in vals):
                stacked = torch.stack(vals)                      # a new tensor: one launch
                logs.append({k: stacked[i] for i, k in enumerate(keys)})
            else:
                logs.append({k: v.clone() for k, v in zip(keys, vals)})
            pred = to['pred']
        if self.training:
            # multi-tensor (foreach) arithmetic throughout: ~100 tensors per list, one or two launches per list
            scaled = lambda gs, w: list(gs) if isinstance(w, float) and w == 1.0 else list(torch._foreach_mul(list(gs), w))
            suffix = None      # sum_{s > t} w_s G_s  over routed tensors
            for t in range(self.S, -1, -1):
                if t in self.target_graphs:
                    to = self.target_graphs[t][1]
                    w = weights[t]
                    gs = to['g_routed']
                    if suffix is None:
                        suffix = scaled(gs, w) if not (isinstance(w, float) and w == 1.0) else hip_ops.mt_clone(gs)
                    else:
                        torch._foreach_add_(suffix, scaled(gs, w))
                    own = [(k, g) for k, g in zip(self.unrouted, to['g_own']) if g is not None]
                    if own:
                        accum.add_params([k for k, _ in own], scaled([g for _, g in own], w))
                if t > 0 and self.learn_lr and suffix is not None:
                    accum.add_lr_grads(self, t - 1, suffix)
            if suffix is not None and gamma is not None:
                # identity chain W_S -> ... -> W_0 = gamma * theta
                g_gamma, g_theta = hip_ops.mt_scale_grads(gamma, [self.theta[k] for k in self.routed], suffix)
                accum.add_params(self.routed, g_theta, owned=True)
                if gamma.requires_grad:
                    att = [p for p in list(sysm.attenuator.parameters()) + [sysm.gamma_mult] if p.requires_grad]
                    accum.add_extra(att, torch.autograd.grad(gamma, att, g_gamma, allow_unused=True))
            elif suffix is not None:
                # identity chain W_S -> ... -> W_0 = theta broadcast over the tasks: the task axis is summed
                accum.add_params(self.routed, suffix if T == 1 else [x.sum(0) for x in suffix], owned=True)
        if T == 1:
            return task_loss.reshape(1), pred.clone(), [logs]
        per_task_logs = [[{k: v[t] for k, v in parts.items()} for parts in logs] for t in range(T)]
        return task_loss, pred.clone(), per_task_logs


class OuterGradAccumulator:
    """Sum over tasks of the manually assembled first-order outer gradients (scaled by 1/B at the end)."""

    def __init__(self, system, theta):
        self.sys = system
        self.theta = theta      # inner-loop key -> nn.Parameter
        self.param = {}         # inner-loop key -> tensor
        self.lr = {}            # lr key -> tensor shaped like the lr parameter (Meta-SGD: element-wise rates)
        self.lr_rows = None     # LSLR: [inner step, routed tensor] scalars
        self.lr_keys = None
        self.extra = {}         # any other trainable parameter (L2F: attenuator, gamma_mult) -> summed gradient

    def add_params(self, keys, grads, owned=False):
        """param[k] += g.  The first contribution of a key is copied -- `grads` may be static graph outputs -- unless the caller hands
        over tensors nobody else holds (`owned`: the suffix sums, which are clones or new sums already)."""
        have = [(self.param[k], g) for k, g in zip(keys, grads) if k in self.param]
        if have:
            torch._foreach_add_([a for a, _ in have], [g for _, g in have])
        fresh = [(k, g) for k, g in zip(keys, grads) if k not in self.param]
        if fresh:
            for (k, g), c in zip(fresh, [g for _, g in fresh] if owned else hip_ops.mt_clone([g for _, g in fresh])):
                self.param[k] = c

    def add_lr_grads(self, gl, t, suffix):
        """dL/d lr_t = <suffix, dir_t> (scalar per tensor for LSLR, element-wise for Meta-SGD)."""
        rule = gl.rule
        dirs = gl.step_out[t]['dir']
        scale = -1.0 if gl.rule_id == _hip.RULE_SGD else 1.0
        lib = _hip.lib()
        n = len(gl.routed)
        numel = [x.numel() for x in suffix]
        gos = [x.contiguous() for x in suffix]
        if rule.lr_mode == _hip.LR_SCALAR:
            dst = torch.zeros(n, dtype=torch.float32, device=gos[0].device)
            outs = [dst[i] for i in range(n)]
        else:
            outs = [torch.empty_like(x) for x in gos]
        args = (rule.lr_mode, n, _hip.ptr_array(gos), _hip.ptr_array(dirs), _hip.ptr_array(outs), _hip.i64_array(numel),
                scale, _hip.current_stream())
        _hip.launch("mt_update_bwd", lambda: _hip.check(lib.savfi_mt_update_bwd_f32(*args), "savfi_mt_update_bwd_f32"))
        lr_keys = [k.replace(".", "-") for k in gl.routed]
        if rule.lr_mode == _hip.LR_SCALAR:
            # one row of per-tensor scalars per inner step; scattered into the lr parameters' .grad by install()
            if self.lr_rows is None:
                steps = max(rule.names_learning_rates_dict[lk].numel() for lk in lr_keys)
                self.lr_rows = torch.zeros(steps, n, dtype=torch.float32, device=dst.device)
                self.lr_keys = lr_keys
            self.lr_rows[t].add_(dst)
        else:
            if gl.T > 1:             # stacked over tasks, ONE rate table: add the tasks
                outs = [o.sum(0) for o in outs]
            live = [(lk, o) for lk, o in zip(lr_keys, outs) if rule.names_learning_rates_dict[lk].requires_grad]
            self.add_lr_tensors([lk for lk, _ in live], [o for _, o in live])

    def add_lr_tensors(self, keys, grads):
        have = [(self.lr[k], g) for k, g in zip(keys, grads) if k in self.lr]
        if have:
            torch._foreach_add_([a for a, _ in have], [g for _, g in have])
        for k, g in zip(keys, grads):
            if k not in self.lr:
                self.lr[k] = g.clone()

    def add_extra(self, params, grads):
        for p, g in zip(params, grads):
            if g is None:
                continue
            if p in self.extra:
                self.extra[p].add_(g)
            else:
                self.extra[p] = g.clone()

    def tensors(self):
        return (list(self.param.values()) + list(self.lr.values()) + ([self.lr_rows] if self.lr_rows is not None else [])
                + list(self.extra.values()))

    def merge(self, other):
        """Add another accumulator's sums (tasks adapted on another stream; the caller has joined the streams)."""
        if other.param:
            self.add_params(list(other.param.keys()), list(other.param.values()))
        if other.lr_rows is not None:
            if self.lr_rows is None:
                self.lr_rows, self.lr_keys = other.lr_rows.clone(), other.lr_keys
            else:
                self.lr_rows.add_(other.lr_rows)
        if other.lr:
            self.add_lr_tensors(list(other.lr.keys()), list(other.lr.values()))
        if other.extra:
            self.add_extra(list(other.extra.keys()), list(other.extra.values()))

    def install(self, num_tasks, into=None):
        """Write the accumulated gradients (mean over the GLOBAL meta-batch) into .grad.  `into` = {param: preassigned .grad
        view} (task parallelism: the views of the all-reduce bucket, TaskParallel.prepare_gradients): the sums are copied there
        instead of re-binding .grad; returns the parameters that received a gradient."""
        inv = 1.0 / float(num_tasks)
        rates = self.sys.inner_loop_optimizer.names_learning_rates_dict
        pairs = []
        if self.lr_rows is not None:
            cols = self.lr_rows.t().mul(inv).contiguous()            # [tensor, step]
            for i, lk in enumerate(self.lr_keys):
                p = rates[lk]
                if p.requires_grad:
                    pairs.append((p, cols[i, :p.numel()].reshape(p.shape)))
        if self.param:
            torch._foreach_mul_(list(self.param.values()), inv)
        pairs += [(self.theta[k], g) for k, g in self.param.items()]
        if self.lr:
            torch._foreach_mul_(list(self.lr.values()), inv)
        pairs += [(rates[lk], g) for lk, g in self.lr.items()]
        if self.extra:
            torch._foreach_mul_(list(self.extra.values()), inv)
        pairs += list(self.extra.items())
        if into is None:
            for p, g in pairs:
                p.grad = g
        elif pairs:
            hip_ops.mt_copy([into[p] for p, _ in pairs], [g.view_as(into[p]) for p, g in pairs])
        return [p for p, _ in pairs]
