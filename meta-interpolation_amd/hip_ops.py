"""torch.autograd wrappers over the non-sepconv entry points of libsavfi_hip.so.

Every function here launches a hand-written gfx950 kernel through the C ABI (include/savfi_hip.h)
on torch's current stream.  Device tensors only -- there is no CPU or eager-PyTorch fallback.
"""
import functools
import ctypes
import os
import threading

import torch
from torch.autograd.function import once_differentiable

from . import _hip

# Set by the meta system for the duration of a forward with --second_order: ops whose hand-written backward
# is not itself differentiable switch to composed device ops (losses) or refuse (voxel warp) instead of
# silently dropping second-order terms.  (The sepconv op does drop them, exactly like the reference:
# SURVEY.md section 0, fact 9.)
# Per THREAD (tasks may be adapted on concurrent threads; two systems in one process must not flip each other's ops):
# worker threads get the caller's value through meta_learning_system._run_tasks.
_PASS = threading.local()


def set_double_backward(value):
    _PASS.double_backward = bool(value)


def double_backward():
    return getattr(_PASS, 'double_backward', False)


# --------------------------------------------------------------------------------------------
# VoxelFlow warp + blend            (reference: voxelflow/core/models/voxel_flow.py:471-509)
# --------------------------------------------------------------------------------------------
class _VoxelWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, frames, x3):
        _hip.require_cuda(frames, x3)
        B, six, H, W = frames.shape
        assert six == 6 and x3.shape == (B, 3, H, W)
        out = torch.empty((B, 3, H, W), dtype=frames.dtype, device=frames.device)
        lib = _hip.lib()
        _hip.launch("voxelwarp_fwd", lambda: _hip.check(lib.savfi_voxelwarp_fwd_f32(
            frames.data_ptr(), x3.data_ptr(), out.data_ptr(), B, H, W, _hip.current_stream()),
            "savfi_voxelwarp_fwd_f32"), nbytes=4 * 12 * B * H * W)
        ctx.save_for_backward(frames, x3)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gO):
        frames, x3 = ctx.saved_tensors
        B, _, H, W = frames.shape
        gO = gO.contiguous()
        need_f, need_x = ctx.needs_input_grad
        g_x3 = torch.empty_like(x3)
        g_fr = torch.zeros_like(frames) if need_f else None
        lib = _hip.lib()
        _hip.launch("voxelwarp_bwd", lambda: _hip.check(lib.savfi_voxelwarp_bwd_f32(
            frames.data_ptr(), x3.data_ptr(), gO.data_ptr(), g_x3.data_ptr(),
            None if g_fr is None else g_fr.data_ptr(), B, H, W, _hip.current_stream()),
            "savfi_voxelwarp_bwd_f32"), nbytes=4 * 15 * B * H * W)
        return g_fr, (g_x3 if need_x else None)


def _bilinear_border(img, cx, cy):
    """F.grid_sample(img, stack(cx, cy), bilinear, padding_mode='border', align_corners=True) written with gathers (ATen
    GridSampler.h: unnormalise, clip the coordinate, four corners): differentiable in `img`, `cx`, `cy` to any order --
    ATen's grid_sampler_2d_backward has no derivative, so the fused op and F.grid_sample alike stop at first order."""
    N, C, H, W = img.shape
    ix = ((cx + 1.0) * 0.5 * (W - 1)).clamp(0, W - 1)
    iy = ((cy + 1.0) * 0.5 * (H - 1)).clamp(0, H - 1)
    x0, y0 = ix.detach().floor(), iy.detach().floor()
    fx, fy = (ix - x0).unsqueeze(1), (iy - y0).unsqueeze(1)
    x0i, y0i = x0.long().clamp(0, W - 1), y0.long().clamp(0, H - 1)
    x1i, y1i = (x0i + 1).clamp(max=W - 1), (y0i + 1).clamp(max=H - 1)
    flat = img.reshape(N, C, H * W)

    def corner(yi, xi):
        return flat.gather(2, (yi * W + xi).reshape(N, 1, H * W).expand(N, C, H * W)).reshape(N, C, H, W)

    top = corner(y0i, x0i) * (1 - fx) + corner(y0i, x1i) * fx
    bot = corner(y1i, x0i) * (1 - fx) + corner(y1i, x1i) * fx
    return top * (1 - fy) + bot * fy


def _voxel_warp_composed(frames, x3):
    """The VoxelFlow tail (voxelflow/core/models/voxel_flow.py:471-507, syn_type 'inter') from differentiable ATen ops: what
    --second_order takes (set_double_backward(True)), like pixel shuffle / up-sampling / the losses."""
    B, _, H, W = frames.shape
    gx = torch.linspace(-1.0, 1.0, W, device=frames.device, dtype=frames.dtype).view(1, 1, W).expand(B, H, W)
    gy = torch.linspace(-1.0, 1.0, H, device=frames.device, dtype=frames.dtype).view(1, H, 1).expand(B, H, W)
    fx, fy = 0.5 * x3[:, 0], 0.5 * x3[:, 1]
    o1 = _bilinear_border(frames[:, 0:3], gx - fx, gy - fy)
    o2 = _bilinear_border(frames[:, 3:6], gx + fx, gy + fy)
    mask = (0.5 * (1.0 + x3[:, 2:3])).expand(B, 3, H, W)
    return mask * o1 + (1.0 - mask) * o2


def voxel_warp_blend(frames, x3):
    """frames [B,6,H,W] (I0|I1), x3 [B,3,H,W] = tanh(conv4) -> interpolated frame [B,3,H,W]."""
    if double_backward():
        return _voxel_warp_composed(frames, x3)
    return _VoxelWarp.apply(frames.contiguous(), x3.contiguous())


# --------------------------------------------------------------------------------------------
# 2x2 average pooling          (reference: sepconv/model.py:176-187, rrin/unet.py:146, superslomo/model.py:66)
# --------------------------------------------------------------------------------------------
class _AvgPool2x2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _hip.require_cuda(x)
        N, C, H, W = x.shape
        out = torch.empty((N, C, H // 2, W // 2), dtype=x.dtype, device=x.device)
        lib = _hip.lib()
        _hip.launch("avgpool2x2_fwd", lambda: _hip.check(lib.savfi_avgpool2x2_fwd_f32(
            x.data_ptr(), out.data_ptr(), N * C, H, W, _hip.current_stream()), "savfi_avgpool2x2_fwd_f32"),
            nbytes=4 * N * C * (H * W + (H // 2) * (W // 2)))
        ctx.hw = (H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        # linear op: its adjoint goes through a Function too, so double-backward keeps working
        return _AvgPool2x2Adjoint.apply(g, ctx.hw)


class _AvgPool2x2Adjoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, hw):
        g = g.contiguous()
        _hip.require_cuda(g)
        N, C = g.shape[:2]
        H, W = hw
        gin = torch.empty((N, C, H, W), dtype=g.dtype, device=g.device)
        lib = _hip.lib()
        _hip.launch("avgpool2x2_bwd", lambda: _hip.check(lib.savfi_avgpool2x2_bwd_f32(
            g.data_ptr(), gin.data_ptr(), N * C, H, W, _hip.current_stream()), "savfi_avgpool2x2_bwd_f32"),
            nbytes=4 * N * C * (H * W + (H // 2) * (W // 2)))
        return gin

    @staticmethod
    def backward(ctx, gg):
        return _AvgPool2x2.apply(gg.contiguous()), None


class _AvgPool2x2AndSkip(torch.autograd.Function):
    """(pool(x), x): an encoder block's activated output feeds the pooling and a skip connection.  backward: ONE pass
    gin = (pool^T(g_pool) + g_skip) * (x > 0 ? 1 : in_slope) -- the pooling's adjoint, autograd's accumulation of the two consumers'
    cotangents and the producer's deferred activation derivative (conv_bias_act `defer`).  First-order passes only."""

    @staticmethod
    def forward(ctx, x, in_slope):
        _hip.require_cuda(x)
        N, C, H, W = x.shape
        out = torch.empty((N, C, H // 2, W // 2), dtype=x.dtype, device=x.device)
        lib = _hip.lib()
        _hip.launch("avgpool2x2_fwd", lambda: _hip.check(lib.savfi_avgpool2x2_fwd_f32(
            x.data_ptr(), out.data_ptr(), N * C, H, W, _hip.current_stream()), "savfi_avgpool2x2_fwd_f32"),
            nbytes=4 * N * C * (H * W + (H // 2) * (W // 2)))
        ctx.in_slope = in_slope
        ctx.save_for_backward(x)
        ctx.set_materialize_grads(False)
        return out, x.view_as(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_pool, g_skip):
        if g_pool is None and g_skip is None:
            return None, None
        x, = ctx.saved_tensors
        N, C, H, W = x.shape
        g_pool = None if g_pool is None else g_pool.contiguous()
        g_skip = None if g_skip is None else g_skip.contiguous()
        gin = torch.empty_like(x)
        lib = _hip.lib()
        _hip.launch("avgpool2x2_bwd", lambda: _hip.check(lib.savfi_avgpool2x2_bwd_fused_f32(
            None if g_pool is None else g_pool.data_ptr(), None if g_skip is None else g_skip.data_ptr(),
            None if ctx.in_slope is None else x.data_ptr(), 1.0 if ctx.in_slope is None else float(ctx.in_slope), gin.data_ptr(),
            N * C, H, W, _hip.current_stream()), "savfi_avgpool2x2_bwd_fused_f32"),
            nbytes=4 * N * C * (H * W * (2 + (g_skip is not None)) + (H // 2) * (W // 2)))
        return gin, None


def avg_pool2x2_and_skip(x, in_slope=None):
    """(avg_pool2x2(x), x) with ONE element-wise pass in backward (see _AvgPool2x2AndSkip); in_slope: x is the activated output of a
    fused convolution that left its activation derivative to this op.  GPU float32 [N,C,H>=2,W>=2] tensors in first-order passes; anything
    else takes the plain ops (with the derivative as an identity node)."""
    if (x.is_cuda and x.dim() == 4 and x.shape[2] >= 2 and x.shape[3] >= 2 and x.dtype == torch.float32 and not double_backward()
            and torch.is_grad_enabled()):
        return _AvgPool2x2AndSkip.apply(x.contiguous(), None if in_slope is None else float(in_slope))
    if in_slope is not None:
        x = mask_grad(x, in_slope)
    return avg_pool2x2(x), x


def avg_pool2x2(x):
    """F.avg_pool2d(x, 2) == nn.AvgPool2d(2, 2): [N,C,H,W] -> [N,C,H//2,W//2]; plain ATen on CPU tensors (host-logic tests)."""
    if not x.is_cuda or x.dim() != 4 or x.shape[2] < 2 or x.shape[3] < 2 or x.dtype != torch.float32:
        return torch.nn.functional.avg_pool2d(x, 2)
    return _AvgPool2x2.apply(x.contiguous())


class AvgPool2x2(torch.nn.Module):
    """Parameter-free stand-in for torch.nn.AvgPool2d(kernel_size=2, stride=2)."""

    def forward(self, x):
        return avg_pool2x2(x)

    def extra_repr(self):
        return 'kernel_size=2, stride=2'


# --------------------------------------------------------------------------------------------
# Backward warp by a pixel-unit flow    (reference: superslomo/model.py:231-307, rrin/model.py:8-20)
# --------------------------------------------------------------------------------------------
class _FlowWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, flow):
        _hip.require_cuda(img, flow)
        N, C, H, W = img.shape
        assert flow.shape == (N, 2, H, W), (img.shape, flow.shape)
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("flow_warp differentiates w.r.t. the flow only: on the reference's path the warped "
                                      "images are network inputs (superslomo/model.py:600-624, rrin/model.py:99-100)")
        out = torch.empty_like(img)
        lib = _hip.lib()
        _hip.launch("flowwarp_fwd", lambda: _hip.check(lib.savfi_flowwarp_fwd_f32(
            img.data_ptr(), flow.data_ptr(), out.data_ptr(), N, C, H, W, _hip.current_stream()),
            "savfi_flowwarp_fwd_f32"), nbytes=4 * N * H * W * (2 * C + 2))
        ctx.save_for_backward(img, flow)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        img, flow = ctx.saved_tensors
        N, C, H, W = img.shape
        gout = gout.contiguous()
        gflow = torch.empty_like(flow)
        lib = _hip.lib()
        _hip.launch("flowwarp_bwd", lambda: _hip.check(lib.savfi_flowwarp_bwd_f32(
            img.data_ptr(), flow.data_ptr(), gout.data_ptr(), gflow.data_ptr(), N, C, H, W, _hip.current_stream()),
            "savfi_flowwarp_bwd_f32"), nbytes=4 * N * H * W * (2 * C + 4))
        return None, gflow


def flow_warp(img, flow):
    """img [N,C,H,W] sampled at (x + u - 0.5, y + v - 0.5), flow = (u, v) [N,2,H,W] in pixels: exactly what
    backWarp / warp of the reference compute (their normalisation under grid_sample's align_corners=False)."""
    return _FlowWarp.apply(img.contiguous(), flow.contiguous())


# --------------------------------------------------------------------------------------------
# Pixel (un)shuffle                                   (reference: model_utils.py:202-217)
# --------------------------------------------------------------------------------------------
def _launch_shuffle(x, r, down):
    _hip.require_cuda(x)
    B, C, H, W = x.shape
    lib = _hip.lib()
    if down:
        assert H % r == 0 and W % r == 0
        out = torch.empty((B, C * r * r, H // r, W // r), dtype=x.dtype, device=x.device)
        _hip.launch("pixel_unshuffle", lambda: _hip.check(lib.savfi_pixel_unshuffle_f32(
            x.data_ptr(), out.data_ptr(), B, C, H, W, r, _hip.current_stream()), "savfi_pixel_unshuffle_f32"), nbytes=8 * x.numel())
    else:
        assert C % (r * r) == 0
        out = torch.empty((B, C // (r * r), H * r, W * r), dtype=x.dtype, device=x.device)
        _hip.launch("pixel_shuffle", lambda: _hip.check(lib.savfi_pixel_shuffle_f32(
            x.data_ptr(), out.data_ptr(), B, C, H, W, r, _hip.current_stream()), "savfi_pixel_shuffle_f32"), nbytes=8 * x.numel())
    return out


class _PixelShuffle(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, r, down):
        ctx.r, ctx.down = r, down
        return _launch_shuffle(x.contiguous(), r, down)

    @staticmethod
    def backward(ctx, g):
        # the permutation's adjoint is its inverse; applied through the Function so that it stays
        # differentiable (second-order MAML through CAIN)
        return _PixelShuffle.apply(g, ctx.r, not ctx.down), None, None


def pixel_shuffle(input, scale_factor):
    """Same call surface as the reference's pixel_shuffle: scale<1 packs space into channels
    (block 1/scale), scale>=1 unpacks."""
    if scale_factor >= 1:
        return _PixelShuffle.apply(input, int(scale_factor), False)
    return _PixelShuffle.apply(input, int(round(1 / scale_factor)), True)


# --------------------------------------------------------------------------------------------
# Per-plane mean removal                              (reference: model_utils.py:11-15 sub_mean)
# --------------------------------------------------------------------------------------------
class _SubMean(torch.autograd.Function):
    """x [N,C,H,W] -> (x - mean_hw(x), mean_hw(x) [N,C,1,1]) on savfi_sub_mean_f32: two launches that add in a fixed order and need
    no cleared memory.  ATen's mean switches to several workgroups per output plus a hipMemsetAsync'ed semaphore array on large
    frames, and a memset node of a captured hipGraph only clears in the first replay on ROCm 7.2 (csrc/submean.hip)."""

    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        _hip.require_cuda(x)
        N, C, H, W = x.shape
        ctx.hw, ctx.shape = H * W, tuple(x.shape)
        out = torch.empty_like(x)
        mean = torch.empty((N, C, 1, 1), dtype=x.dtype, device=x.device)
        ws = torch.empty(_workspace_floats("savfi_sub_mean_workspace_floats", N * C, H * W), dtype=x.dtype, device=x.device)
        lib = _hip.lib()
        _hip.launch("sub_mean", lambda: _hip.check(lib.savfi_sub_mean_f32(
            x.data_ptr(), out.data_ptr(), mean.data_ptr(), ws.data_ptr(), N * C, H * W, _hip.current_stream()), "savfi_sub_mean_f32"),
            nbytes=12 * x.numel())
        return out, mean

    @staticmethod
    def backward(ctx, g_out, g_mean):
        # out = x - m(x), mean = m(x), m linear and self-adjoint up to 1/hw: gx = g_out - m(g_out) + g_mean / hw  (differentiable:
        # the same Function and ATen arithmetic)
        gx = None
        if g_out is not None:
            gx = _SubMean.apply(g_out)[0]
        if g_mean is not None:
            share = (g_mean / ctx.hw).expand(ctx.shape)
            gx = share if gx is None else gx + share
        return gx


def sub_mean(x):
    """(x - mean, mean): the per-channel spatial mean removed (reference model_utils.py:11-15)."""
    return _SubMean.apply(x)


# --------------------------------------------------------------------------------------------
# Fused multi-tensor inner-loop update            (reference: inner_loop_optimizers.py)
# --------------------------------------------------------------------------------------------
class _MtUpdate(torch.autograd.Function):
    """(w_1..w_n, lr_1..lr_n) -> (w'_1..w'_n); g, moments and hyper-parameters ride in `spec`.

    First-order MAML semantics: the gradients g are constants.  d w'/d w = I, and d w'/d lr is the
    per-element update direction, saved by the forward kernel when some lr requires grad.
    """

    @staticmethod
    def forward(ctx, spec, *tensors):
        ctx.set_materialize_grads(False)   # unused outputs (e.g. SepConv's subnet copies) stay None
        n = spec["n"]
        ws, lrs = tensors[:n], tensors[n:]
        gs = spec["grads"]
        rule, lr_mode = spec["rule"], spec["lr_mode"]
        outs = [torch.empty_like(w) for w in ws]
        need_lr = any(ctx.needs_input_grad[1 + n + i] for i in range(n))
        save_dir = need_lr and rule != _hip.RULE_SGD
        coefs = [torch.empty_like(w) for w in ws] if save_dir else None
        ms, ss = spec.get("m"), spec.get("s")
        _launch_mt_update(rule, lr_mode, ws, gs, lrs, ms, ss, outs, coefs, spec.get("bc1"), spec.get("sqrt_bc2"),
                          spec["beta1"], spec["beta2"], spec["eps"])
        numel = [w.numel() for w in ws]
        ctx.n, ctx.lr_mode, ctx.numel = n, lr_mode, numel
        ctx.lr_shapes = [lr.shape for lr in lrs]
        ctx.w_shapes = [w.shape for w in ws]
        if need_lr:
            ctx.dirs = coefs if save_dir else list(gs)
            ctx.dir_scale = 1.0 if save_dir else -1.0
        else:
            ctx.dirs = None
        return tuple(outs)

    @staticmethod
    def backward(ctx, *g_outs):
        n = ctx.n
        g_ws = [g if ctx.needs_input_grad[1 + i] else None for i, g in enumerate(g_outs)]
        g_lrs = [None] * n
        if ctx.dirs is not None:
            idx = [i for i in range(n) if ctx.needs_input_grad[1 + n + i] and g_outs[i] is not None]
            if idx:
                gos = [g_outs[i].contiguous() for i in idx]
                dirs = [ctx.dirs[i] for i in idx]
                dev = gos[0].device
                if ctx.lr_mode == _hip.LR_SCALAR:      # (one zero fill for all of them: 0-dim views of one buffer)
                    dst = list(torch.zeros(len(idx), dtype=torch.float32, device=dev).unbind(0))
                else:
                    dst = [torch.empty_like(go) for go in gos]
                lib = _hip.lib()
                numel = [ctx.numel[i] for i in idx]
                args = (ctx.lr_mode, len(idx), _hip.ptr_array(gos), _hip.ptr_array(dirs), _hip.ptr_array(dst),
                        _hip.i64_array(numel), ctx.dir_scale, _hip.current_stream())
                _hip.launch("mt_update_bwd", lambda: _hip.check(lib.savfi_mt_update_bwd_f32(*args),
                                                                 "savfi_mt_update_bwd_f32"))
                for j, i in enumerate(idx):
                    if ctx.lr_mode == _hip.LR_ELEMENT and tuple(ctx.w_shapes[i]) != tuple(ctx.lr_shapes[i]):
                        g_lrs[i] = dst[j].sum(0)         # weights stacked over tasks, ONE element-wise lr table: add the tasks
                    else:
                        g_lrs[i] = dst[j].reshape(ctx.lr_shapes[i])
        return (None, *g_ws, *g_lrs)


def _launch_mt_update(rule, lr_mode, ws, gs, lrs, ms, ss, outs, coefs, bc1, sqrt_bc2, beta1, beta2, eps):
    """One savfi_mt_update_f32 call.  Weights stacked over tasks ([T, *shape], tasks adapted in lockstep) with an element-wise
    learning-rate table of shape `shape` (Meta-SGD) go in as T entries, one per task slice, that share the lr pointer; with a
    scalar lr (LSLR) a stacked tensor is just a tensor of T x numel elements."""
    if lr_mode == _hip.LR_ELEMENT and any(w.shape != lr.shape for w, lr in zip(ws, lrs)):
        def cut(ts):
            if ts is None:
                return None
            out = []
            for t, w, lr in zip(ts, ws, lrs):
                out.extend([t] if w.shape == lr.shape else list(t.unbind(0)))
            return out
        rep = lambda vals: None if vals is None else [v for v, w, lr in zip(vals, ws, lrs)
                                                      for _ in range(1 if w.shape == lr.shape else w.shape[0])]
        lrs = rep(list(lrs))
        bc1, sqrt_bc2 = rep(bc1), rep(sqrt_bc2)
        ws_, gs, ms, ss, outs, coefs = cut(ws), cut(gs), cut(ms), cut(ss), cut(outs), cut(coefs)
        ws = ws_
    lib = _hip.lib()
    args = (rule, lr_mode, len(ws), _hip.ptr_array(ws), _hip.ptr_array(gs), _hip.ptr_array(lrs),
            _hip.ptr_array(ms) if ms is not None else None, _hip.ptr_array(ss) if ss is not None else None,
            _hip.ptr_array(outs), _hip.ptr_array(coefs) if coefs is not None else None,
            _hip.i64_array([w.numel() for w in ws]),
            _hip.f32_array(bc1) if bc1 is not None else None, _hip.f32_array(sqrt_bc2) if sqrt_bc2 is not None else None,
            beta1, beta2, eps, _hip.current_stream())
    per_el = 12 + (4 if lr_mode == _hip.LR_ELEMENT else 0) + (8 if ms is not None else 0) + (8 if ss is not None else 0) + (4 if coefs is not None else 0)
    _hip.launch("mt_update", lambda: _hip.check(lib.savfi_mt_update_f32(*args), "savfi_mt_update_f32"),
                nbytes=per_el * sum(w.numel() for w in ws))


def mt_update(rule, lr_mode, weights, grads, lrs, m=None, s=None, bc1=None, sqrt_bc2=None,
              beta1=0.9, beta2=0.99, eps=1e-8):
    """Fused update of a list of tensors.  `lrs[i]` is a 0-dim tensor (LR_SCALAR) or a tensor shaped
    like weights[i] (LR_ELEMENT).  m / s are updated in place by the kernel.  Returns new tensors."""
    n = len(weights)
    if n == 0:
        return []
    _hip.require_cuda(*weights, *grads, *lrs)
    if m is not None:
        _hip.require_cuda(*m)
    if s is not None:
        _hip.require_cuda(*s)
    spec = dict(n=n, rule=rule, lr_mode=lr_mode, grads=[g.detach() for g in grads], m=m, s=s, bc1=bc1,
                sqrt_bc2=sqrt_bc2, beta1=beta1, beta2=beta2, eps=eps)
    ws = [w if w.is_contiguous() else w.contiguous() for w in weights]
    outs = list(_MtUpdate.apply(spec, *ws, *lrs))
    filters_after_update(outs)
    return outs


def mt_update_nograd(rule, lr_mode, weights, grads, lrs, m, s, bc1, sqrt_bc2, beta1, beta2, eps, want_coef):
    """The fused update without the autograd wrapper (used inside captured hipGraphs, where the outer gradient
    is assembled by hand): returns (new weights, coef or None) with coef = d w' / d lr per element."""
    outs = [torch.empty_like(w) for w in weights]
    coefs = [torch.empty_like(w) for w in weights] if want_coef else None
    _launch_mt_update(rule, lr_mode, list(weights), list(grads), list(lrs), m, s, outs, coefs, bc1, sqrt_bc2, beta1, beta2, eps)
    filters_after_update(outs)
    return outs, coefs


# --------------------------------------------------------------------------------------------
# L2F: per-tensor mean of gradients, per-tensor attenuation   (meta_learning_system.py:249-268)
# --------------------------------------------------------------------------------------------
def mt_mean(tensors):
    """[t_1..t_n] -> float32[n] of per-tensor means (no autograd: inputs are first-order grads)."""
    ts = [t.detach().contiguous() for t in tensors]
    _hip.require_cuda(*ts)
    out = torch.zeros(len(ts), dtype=torch.float32, device=ts[0].device)
    lib = _hip.lib()
    args = (len(ts), _hip.ptr_array(ts), _hip.i64_array([t.numel() for t in ts]), out.data_ptr(),
            _hip.current_stream())
    _hip.launch("mt_mean", lambda: _hip.check(lib.savfi_mt_mean_f32(*args), "savfi_mt_mean_f32"))
    return out


class _MtScale(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gamma, *ws):
        ctx.set_materialize_grads(False)
        _hip.require_cuda(gamma, *ws)
        outs = [torch.empty_like(w) for w in ws]
        lib = _hip.lib()
        numel = [w.numel() for w in ws]
        args = (len(ws), _hip.ptr_array(ws), gamma.data_ptr(), _hip.ptr_array(outs), _hip.i64_array(numel),
                _hip.current_stream())
        _hip.launch("mt_scale", lambda: _hip.check(lib.savfi_mt_scale_f32(*args), "savfi_mt_scale_f32"))
        ctx.save_for_backward(gamma, *ws)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *g_outs):
        gamma, *ws = ctx.saved_tensors
        n = len(ws)
        live = [i for i in range(n) if g_outs[i] is not None]
        g_ws = [None] * n
        g_gamma = torch.zeros_like(gamma) if ctx.needs_input_grad[0] else None
        if live:
            # gamma / g_gamma are indexed by tensor position: run the live tensors through a compacted
            # gamma vector and scatter the per-tensor reductions back
            gos = [g_outs[i].contiguous() for i in live]
            wl = [ws[i] for i in live]
            gw_l = [torch.empty_like(ws[i]) if ctx.needs_input_grad[1 + i] else None for i in live]
            full = len(live) == n
            gam_l = gamma if full else gamma[live].contiguous()
            gg_l = None if g_gamma is None else (g_gamma if full else torch.zeros_like(gam_l))
            lib = _hip.lib()
            args = (len(live), _hip.ptr_array(gos), _hip.ptr_array(wl), gam_l.data_ptr(), _hip.ptr_array(gw_l),
                    None if gg_l is None else gg_l.data_ptr(), _hip.i64_array([w.numel() for w in wl]),
                    _hip.current_stream())
            _hip.launch("mt_scale_bwd", lambda: _hip.check(lib.savfi_mt_scale_bwd_f32(*args),
                                                            "savfi_mt_scale_bwd_f32"))
            for j, i in enumerate(live):
                g_ws[i] = gw_l[j]
            if g_gamma is not None and not full:
                g_gamma[live] = gg_l
        return (g_gamma, *g_ws)


def mt_scale_grads(gamma, weights, g_outs):
    """The backward of mt_scale without autograd (hipGraph loop, where the outer gradient is assembled by hand):
    returns (g_gamma float32[n] = <g_outs[i], weights[i]>, [gamma[i] * g_outs[i]])."""
    ws = [w.detach().contiguous() for w in weights]
    gos = [g.contiguous() for g in g_outs]
    gamma = gamma.detach().contiguous()
    _hip.require_cuda(gamma, *ws, *gos)
    gws = [torch.empty_like(w) for w in ws]
    gg = torch.zeros_like(gamma)
    lib = _hip.lib()
    args = (len(ws), _hip.ptr_array(gos), _hip.ptr_array(ws), gamma.data_ptr(), _hip.ptr_array(gws), gg.data_ptr(),
            _hip.i64_array([w.numel() for w in ws]), _hip.current_stream())
    _hip.launch("mt_scale_bwd", lambda: _hip.check(lib.savfi_mt_scale_bwd_f32(*args), "savfi_mt_scale_bwd_f32"))
    return gg, gws


def mt_scale(gamma, weights):
    """gamma float32[n] (device), weights list of n tensors -> [gamma[i] * weights[i]]."""
    ws = [w if w.is_contiguous() else w.contiguous() for w in weights]
    outs = list(_MtScale.apply(gamma.contiguous(), *ws))
    # L2F's attenuated weights are born together like an update's fast weights: their filters in one launch per kind (config C5: 254
    # single-layer transforms of ~9 us per meta-iteration otherwise); defined further down, with the plan it keeps per list of shapes
    filters_after_update([o.detach() for o in outs])
    return outs


_ONES = {}


def _ones(n, device):
    t = _ONES.get(device)
    if t is None or t.numel() < n:
        t = _ONES[device] = torch.ones(max(int(n), 1024), dtype=torch.float32, device=device)
    return t


def mt_scale_into(gamma, weights, outs):
    """outs[i] <- gamma[i] * weights[i] without autograd, into tensors the caller owns (the static W_0 buffers of the hipGraph loop)."""
    ws = [w.detach() if w.is_contiguous() else w.detach().contiguous() for w in weights]
    outs = [o.detach() for o in outs]
    gamma = gamma.detach().contiguous()
    _hip.require_cuda(gamma, *ws, *outs)
    assert len(ws) == len(outs) and all(o.is_contiguous() and o.numel() == w.numel() and o.dtype == w.dtype for o, w in zip(outs, ws))
    lib = _hip.lib()
    args = (len(ws), _hip.ptr_array(ws), gamma.data_ptr(), _hip.ptr_array(outs), _hip.i64_array([w.numel() for w in ws]),
            _hip.current_stream())
    _hip.launch("mt_scale", lambda: _hip.check(lib.savfi_mt_scale_f32(*args), "savfi_mt_scale_f32"))


def mt_copy(dsts, srcs):
    """dsts[i] <- srcs[i] for lists of float32 CUDA tensors of equal sizes: ceil(n / 48) launches of the multi-tensor scale kernel
    with gamma = 1 (x * 1.0f is x, bit for bit) where torch._foreach_copy_ issues one device memcpy per tensor -- 494 of them per
    task for CAIN, twice per meta-iteration of the hipGraph loop (theta -> W_0, first gradient -> accumulator): ~1000 host-bound
    launches of a 26 ms iteration at 64 x 64."""
    dsts, srcs = list(dsts), list(srcs)
    if not dsts:
        return
    plain = all(d.is_cuda and d.dtype == torch.float32 and s.dtype == torch.float32 and s.device == d.device and d.is_contiguous()
                and d.numel() == s.numel() and d.numel() > 0 for d, s in zip(dsts, srcs))
    if not plain:
        with torch.no_grad():
            torch._foreach_copy_(dsts, [s.view_as(d) if s.shape != d.shape and s.numel() == d.numel() else s for d, s in zip(dsts, srcs)])
        return
    mt_scale_into(_ones(len(dsts), dsts[0].device), srcs, dsts)


class MtCopy:
    """mt_copy between two FIXED lists (the parameters and the static W_0 buffers of a hipGraph set: neither is ever re-allocated):
    the pointer tables are built once -- 2 x 494 ctypes stores per call for CAIN otherwise -- and run() re-checks the addresses."""

    def __init__(self, dsts, srcs):
        # the LIVE objects are kept (a `.detach()` alias would keep pointing at the old storage after `p.data = ...` --
        # module.to() / .float() -- and the address check below could never fail)
        self.dsts, self.srcs = list(dsts), list(srcs)
        self.plain = bool(self.dsts) and all(
            d.is_cuda and d.dtype == torch.float32 and s.dtype == torch.float32 and s.device == d.device and d.is_contiguous()
            and s.is_contiguous() and d.numel() == s.numel() and d.numel() > 0 for d, s in zip(self.dsts, self.srcs))
        if self.plain:
            self.ptrs = [t.data_ptr() for t in self.dsts + self.srcs]
            self.args = (len(self.dsts), _hip.ptr_array(self.srcs), None, _hip.ptr_array(self.dsts),
                         _hip.i64_array([d.numel() for d in self.dsts]))

    def run(self):
        if not self.dsts:
            return
        if not self.plain or any(t.data_ptr() != p for t, p in zip(self.dsts + self.srcs, self.ptrs)):
            with torch.no_grad():
                mt_copy([d.detach() for d in self.dsts], [s.detach() for s in self.srcs])
            return
        n, pw, _, po, numel = self.args
        gamma = _ones(n, self.dsts[0].device)
        lib = _hip.lib()
        _hip.launch("mt_scale", lambda: _hip.check(lib.savfi_mt_scale_f32(n, pw, gamma.data_ptr(), po, numel, _hip.current_stream()),
                                                   "savfi_mt_scale_f32"))


def mt_clone(srcs):
    """[s.clone() for s in srcs] through mt_copy."""
    srcs = list(srcs)
    outs = [torch.empty_like(s, memory_format=torch.contiguous_format) for s in srcs]
    mt_copy(outs, srcs)
    return outs


# --------------------------------------------------------------------------------------------
# Fused L1 / MSE                                                     (loss.py:287-290)
# --------------------------------------------------------------------------------------------
class _L1Mse(torch.autograd.Function):
    """a, b [rows, ...] -> float32[rows] of per-row mean |a-b| (kind 0) / mean (a-b)^2 (kind 1); deterministic."""

    @staticmethod
    def forward(ctx, kind, a, b):
        _hip.require_cuda(a, b)
        assert a.shape == b.shape
        rows = a.shape[0]
        n = a.numel() // rows
        res = torch.empty(rows, dtype=torch.float32, device=a.device)
        scratch = torch.empty(_workspace_floats("savfi_l1_mse_scratch_floats", rows, n), dtype=torch.float32, device=a.device)
        lib = _hip.lib()
        _hip.launch("l1_mse", lambda: _hip.check(lib.savfi_l1_mse_f32(
            kind, a.data_ptr(), b.data_ptr(), res.data_ptr(), scratch.data_ptr(), rows, n, _hip.current_stream()),
            "savfi_l1_mse_f32"))
        ctx.kind = kind
        ctx.save_for_backward(a, b)
        return res

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        ga = torch.empty_like(a)
        rows = a.shape[0]
        lib = _hip.lib()
        _hip.launch("l1_mse_bwd", lambda: _hip.check(lib.savfi_l1_mse_bwd_f32(
            ctx.kind, a.data_ptr(), b.data_ptr(), g.data_ptr(), ga.data_ptr(), rows, a.numel() // rows,
            _hip.current_stream()), "savfi_l1_mse_bwd_f32"))
        gb = -ga if ctx.needs_input_grad[2] else None
        return None, ga, gb


def _loss_rows(kind, a, b):
    a, b = a.contiguous(), b.contiguous()
    return _L1Mse.apply(kind, a.reshape(1, -1) if a.dim() < 2 else a, b.reshape(1, -1) if b.dim() < 2 else b)


def l1_loss(a, b):
    """nn.L1Loss(): mean over everything (loss.py:287)."""
    if double_backward():
        return torch.nn.functional.l1_loss(a, b)
    return _L1Mse.apply(0, a.contiguous().reshape(1, -1), b.contiguous().reshape(1, -1)).reshape(())


def mse_loss(a, b):
    if double_backward():
        return torch.nn.functional.mse_loss(a, b)
    return _L1Mse.apply(1, a.contiguous().reshape(1, -1), b.contiguous().reshape(1, -1)).reshape(())


def l1_loss_per_sample(a, b):
    """[N,...] x [N,...] -> [N]: nn.L1Loss() of every sample on its own (tasks adapted in lockstep), one launch."""
    if double_backward():
        return (a - b).abs().flatten(1).mean(1)
    return _loss_rows(0, a, b)


def mse_loss_per_sample(a, b):
    if double_backward():
        return (a - b).pow(2).flatten(1).mean(1)
    return _loss_rows(1, a, b)


# --------------------------------------------------------------------------------------------
# conv + bias + (leaky) ReLU with fused epilogues   (sepconv/model.py:172-194, model_utils.py:957-990)
# --------------------------------------------------------------------------------------------
# 3x3 / stride 1 convolutions run on savfi_conv3x3_f32 (Winograd on the fp32 matrix cores, bias + activation in its
# epilogue) wherever it beats MIOpen; tiles = N * ceil(Ho/2) * ceil(Wo/2).  Thresholds from tools/conv_bench.py with the
# late round-2 kernel (profiles/r02_conv_bench.jsonl: filter transform included): at N >= 2 it is 1.15-2.1x faster forward and
# 1.1-1.9x for the data gradient on every layer shape of the four plugins down to 24x32 maps (12x16 / 16x16 stay on MIOpen:
# forward 1.27x, data gradient 0.87x, on ~40 us kernels); at N = 1 it wins from 48x64 maps up (1.05-1.9x) and loses on 24x32 (0.86-1.04x).  (Round 1 needed 6000-
# 20000 tiles: the corner-tile tail and the output stage's store stalls, DESIGN.md 4b, weighed most on small maps.)
WINOGRAD_CONV = True            # (routing knobs are module attributes: tools and tests set them, no environment variable is read)
WINO_MIN_TILES_FWD = 700
WINO_MIN_TILES_BWD = 700
WINO_MIN_TILES_FWD_BATCHED = 100      # N >= 2 (support pairs, lockstep batches of shared weights): 16x16 maps and up -- with the filters of all
                                      # layers transformed in one launch per inner step (round 3) the forward is 16.5 vs 31.5 us on CAIN's
                                      # 192 -> 192 layers at 16x16, N = 2 and the data gradient 16.1 vs 22.0 (profiles/r03_convk_bench.jsonl);
                                      # config C1: forward 33.7 -> 38.4 steps/s, data gradient another +3 %.  Single samples at that size
                                      # (64 tiles) measured 1 % slower than MIOpen in the loop: they keep the 700 above
WINO_MIN_TILES_BWD_BATCHED = 100
# The weight gradient of the same layers runs on savfi_conv3x3_wgrad_f32 (NCHW-native MFMA kernel, deterministic) when a
# map has >= 3000 output pixels and >= 16 input channels: 1.2-1.8x faster than MIOpen's igemm kernel + its two layout
# transposes there (tools/wgrad_bench.py, profiles/r01_wgrad_bench.jsonl); the deep 24x32 / 12x16 layers stay on MIOpen.
WGRAD_MIN_PIXELS = 3000
WGRAD_MIN_CI = 16
# Winograd form of the weight gradient (savfi_conv3x3_wgrad_wino_tasks_f32, F(3x3, 2x2)): 1.4-1.6x faster than the direct kernel
# on the large layers and 1.2-1.5x faster than MIOpen (grouped or not) on the deep 24x32 / 12x16 ones once a call carries enough
# work; below ~5 GFLOP the three launches (kernel + two reduction levels) are the cost and the old routing stays.
WGRAD_WINO = True
WGRAD_WINO_MIN_GFLOP = 5.0
WGRAD_WINO_MIN_PIXELS = 192


def _wgrad_wino(N, Ci, Co, Ho, Wo):
    # the Winograd form indexes a sample's channels with 28-bit element offsets (SAVFI_E_TOOBIG beyond): oversize layers fall back
    if Ci * (Ho + 2) * (Wo + 2) >= (1 << 28) or Co * Ho * Wo >= (1 << 28):
        return False
    return WGRAD_WINO and Ho * Wo >= WGRAD_WINO_MIN_PIXELS and 18e-9 * Ci * Co * Ho * Wo * N >= WGRAD_WINO_MIN_GFLOP


# Direct K x K convolution on split-bf16 MFMAs (savfi_convk_*: csrc/convk.hip, csrc/convk_wgrad.hip).  fp32-equivalent arithmetic
# (six bf16 products per fp32 product, fp32 accumulate; as close to fp64 as an fp32 fmaf chain, tools/bf16_split_probe.hip) at
# up to 2.4x the fp32 matrix rate.  Routing (tools/convk_bench.py, profiles/r03_convk_bench.txt):
#   5x5 / 7x7 (VoxelFlow, Super SloMo): always -- forward / data gradient 1.8-3x MIOpen's igemm kernels, weight gradient 1.4-1.7x;
#   3x3: where it beats the Winograd kernel -- layers of >= 64 -> 64 channels in whole 64-channel blocks on maps of >= 700 pixels
#        (220-240 vs 200-208 direct-equivalent TFLOP/s) and the <= 8-channel input layers -- or where a plugin asks for the
#        direct form because it amplifies Winograd rounding (VoxelFlow: `direct=True`).
CONVK = True
CONVK_3X3_MIN_PIXELS = 700


CONVK_WGRAD3_RING_MIN_PIXELS = 3000
CONVK_WGRAD3_RING_SMALL_MIN_PIXELS = 64
CONVK_WGRAD3 = True             # A/B: False = every 3 x 3 weight gradient on the Winograd form
CONVK_WGRAD3_RING = True        # A/B: False = no all-taps kernel (the tap-split kernel's rules alone)


def convk_wgrad_preferred(K, Ci, Co, Ho, Wo, direct=False, N=None):
    """Weight gradient of a stride-1 K x K layer on csrc/convk_wgrad.hip rather than the Winograd / MIOpen forms?  5x5 / 7x7 and
    `direct` layers always; 3x3 where the split-bf16 kernels measured faster than savfi_conv3x3_wgrad_wino:
    * >= 48 -> 48 channels on maps of >= 3000 pixels -- the all-taps kernel on the row ring (round 5; tools/wgrad3_forms_time.py,
      profiles/r05_wgrad3_forms.txt: 160-167 vs 181-184 us on 64 -> 64 @192x256 / 128 -> 128 @96x128, 82 vs 99 on 128 -> 64 @96x128, 57 vs 69 on
      64 -> 64 @96x128 at T = 4 x 2 samples; CAIN 192 -> 192 @96x160: 127 vs 164; the deep 24x32 / 12x16 layers stay on Winograd: 215 vs 180);
    * the wide shallow layers (<= 32 input channels: 89 vs 167 us for 6 -> 32 and 195 vs 252 us for 32 -> 32 at 384 x 512, T = 4)
      and >= 192 output channels on maps of >= 4096 pixels on the tap-split kernel (profiles/r03_convk_bench.jsonl)."""
    if not CONVK or K not in (3, 5, 7):
        return False
    if K != 3 or direct:
        return True
    if not CONVK_WGRAD3:
        return False
    if Ci >= 48 and Co >= 48 and Ho * Wo >= CONVK_WGRAD3_RING_MIN_PIXELS and CONVK_WGRAD3_RING:
        return True
    # ... and the small maps the Winograd form does not take either (N samples; CAIN at 64x64: 192 -> 192 @16x16, where MIOpen's
    # weight gradient is im2col + two GEMMs + col2im, 33 us in four launches against 12 + 5: config C1 45.5 -> 47.1 steps/s)
    if N is not None and Ci >= 48 and Co >= 48 and Ho * Wo >= CONVK_WGRAD3_RING_SMALL_MIN_PIXELS and CONVK_WGRAD3_RING \
            and not _wgrad_wino(N, Ci, Co, Ho, Wo):
        return True
    return (Ci <= 32 and Ho * Wo >= 16384) or (Co >= 192 and Ho * Wo >= 4096)


def _convk_geometry(weight, stride, padding, dilation, groups):
    """(K, pad) if the layer is a square K x K / stride 1 / undilated / ungrouped convolution with symmetric padding the direct
    kernels take, else None."""
    one = lambda v, k: (v == k) if isinstance(v, int) else all(t == k for t in v)
    K = int(weight.shape[-1])
    if K not in (3, 5, 7) or int(weight.shape[-2]) != K or not one(stride, 1) or not one(dilation, 1) or groups != 1:
        return None
    pad = padding if isinstance(padding, int) else (padding[0] if padding[0] == padding[1] else -1)
    if pad < 0 or pad > K - 1:
        return None
    return K, int(pad)


# Winograd F(4x4, 3x3) (csrc/winograd4.h; round 6): the 3x3 layers of at most 512 -> 512 channels, wherever its launch fills the chip --
# measured against the direct split-bf16 kernel on one box (tools/r6/wino4_time.py deep|small convk, profiles/r06_wino4_vs_convk.txt;
# forward / data gradient in us, T = 4 x 2 samples): 64 -> 64 @192x256 95 / 94 vs 142 / 147, 128 -> 128 @96x128 94 / 92 vs 128 / 128,
# 256 -> 256 @48x64 118 / 115 vs 133 / 132, 64 -> 64 @137x236 N = 32 324 / 320 vs 385 / 393, 64 -> 64 @96x128 35 / 34 vs 47 / 46, 128 -> 128
# @48x64 (192 workgroups) 37 / 36 vs 42 / 42; with the reduction split over workgroups (from 256 channels): 512 -> 512 @24x32 138 / 141 vs
# 164 / 160, 256 -> 256 @24x32 46 / 44 vs 57 / 56, 512 -> 512 @12x16 64 / 62 (F(2x2): 53; direct: 107).  Below ~180 workgroups (of 2 per CU x
# 256 CUs = 512 slots) the direct kernel keeps the layer.
WINO4_MIN_WORKGROUPS = 180
WINO4_MIN_PIXELS = 400


@functools.lru_cache(maxsize=None)
def _conv3x3_name(Ci, Co, what):
    """Timer name of a savfi_conv3x3_* launch: 'conv3x3f4_*' for the layers the library runs on its F(4x4) kernel (by channel counts)."""
    f4 = int(_hip.lib().savfi_conv3x3_f4_workgroups(1, int(Ci), int(Co), 8, 8, 1, 0)) > 0
    return ("conv3x3f4_" if f4 else "conv3x3_") + what


@functools.lru_cache(maxsize=4096)
def wino4_workgroups(N, Ci, Co, H, W, pad, mode=0):
    """Workgroups savfi_conv3x3_* would launch on its F(4x4) kernel for this call; 0: the layer's channel counts keep it on F(2x2).
    (Cached per shape: the routing asks once per convolution call, and a 64 x 64 CAIN iteration is host-bound.)"""
    n = int(_hip.lib().savfi_conv3x3_f4_workgroups(int(N), int(Ci), int(Co), int(H), int(W), int(pad), int(mode)))
    return max(n, 0)


def convk_eligible(x, weight, stride, padding, dilation, groups=1, direct=False):
    """Does this convolution (and its data gradient) run on the direct split-bf16 kernel?  weight [Co,Ci,K,K] or [T,Co,Ci,K,K]."""
    if not (CONVK and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return False
    geo = _convk_geometry(weight, stride, padding, dilation, groups)
    if geo is None:
        return False
    K, pad = geo
    H, W = x.shape[2:]
    Ho, Wo = H + 2 * pad - K + 1, W + 2 * pad - K + 1
    Co, Ci = weight.shape[-4], weight.shape[-3]
    if Ho < 1 or Wo < 1 or Ci * H * W >= (1 << 29) or Co * Ho * Wo >= (1 << 29):
        return False
    if K != 3 or direct:
        return True
    if not (Ho * Wo >= CONVK_3X3_MIN_PIXELS and (Ci <= 8 or (Ci >= 64 and Co >= 64 and Co % 64 == 0))):
        return False
    # (the <= 8-channel input layers stay here: an F(4x4) chunk is 8 reduction channels, half of them padding for 6 -> 32)
    return Ci <= 8 or pad > 1 or not WINOGRAD_CONV or wino4_workgroups(int(x.shape[0]), int(Ci), int(Co), int(H), int(W), int(pad)) < WINO4_MIN_WORKGROUPS


def wino_form2(x, weight, pad):
    """Does a layer on the Winograd route run the F(2x2) kernel although its channel counts would put it on F(4x4)?  Yes where the F(4x4)
    launch would not fill the chip (small maps: the 64 x 64 fixtures, config C1): F(2x2) splits its reduction from 128 workgroups down and
    rounds 5x finer (an Adam-type inner rule turns F(4x4)'s rounding into flipped steps of the elements whose gradient is rounding
    noise: CAIN 64 x 64 + Adam).  The filters of such a layer are kind 'wino2' (savfi_conv3x3_*_form_f32 / bit 1 of `mode`)."""
    Co, Ci = weight.shape[-4], weight.shape[-3]
    H, W = int(x.shape[2]), int(x.shape[3])
    n = wino4_workgroups(int(x.shape[0]), int(Ci), int(Co), H, W, int(pad))
    if n <= 0:
        return False
    # ... and on maps below ~400 pixels: a 12 x 16 map is 12 of a workgroup's 32 tiles, and the 36-point filter transform of a deep layer
    # (2.25x F(2x2)'s: 151 MB per 512 -> 512 layer, direction and inner step at T = 4) costs more than the kernel saves there:
    # 512 -> 512 @12x16, T = 4 x 2: 67 + 67 us + 92 us of transforms on F(4x4) against 53 + 52 + 41 on F(2x2)
    return n < WINO4_MIN_WORKGROUPS or (H + 2 * pad - 2) * (W + 2 * pad - 2) < WINO4_MIN_PIXELS


# Packed / transformed filters of a module's OWN parameters are cached per weight version: a first-order meta-iteration
# reads them in every support and target pass (SepConv's Subnets: 11 passes) and only the outer optimizer step changes them.
# The cache is a dict OWNED BY THE MODULE (MetaConv2dLayer passes its own): it dies with the module, so a parameter of a later
# module that happens to reuse the address can never alias an entry.  Fast weights are new tensors every inner step and are
# never cached.  Nothing is looked up or stored while a hipGraph is being captured (a replay must recompute the filters from
# the live weights).
_FILTER_CACHE_PER_MODULE = 6


def _filters(kind, weight, fwd, bwd, cache):
    make = convk_filters if kind == 'convk' else (conv3x3_filters if kind == 'wino' else (lambda w_, f_, b_: conv3x3_filters(w_, f_, b_, f2=True)))
    if cache is None or torch.cuda.is_current_stream_capturing():
        return make(weight, fwd, bwd)
    key = (kind, weight.data_ptr(), weight._version, tuple(weight.shape), weight.device.index, _hip.current_stream())
    hit = cache.get(key)
    if hit is not None and (hit[0] is not None or not fwd) and (hit[1] is not None or not bwd):
        return (hit[0] if fwd else None), (hit[1] if bwd else None)
    pf, pb = make(weight, fwd or (hit is not None and hit[0] is not None), bwd or (hit is not None and hit[1] is not None))
    cache.pop(key, None)
    while len(cache) >= _FILTER_CACHE_PER_MODULE:
        cache.pop(next(iter(cache)))
    cache[key] = (pf, pb)
    return (pf if fwd else None), (pb if bwd else None)


def conv3x3_eligible(x, weight, stride, padding, dilation, groups, backward=False):
    if not (WINOGRAD_CONV and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return False
    one = lambda v, k: (v == k) if isinstance(v, int) else all(t == k for t in v)
    pad = padding if isinstance(padding, int) else (padding[0] if padding[0] == padding[1] else -1)
    if tuple(weight.shape[2:]) != (3, 3) or not one(stride, 1) or not one(dilation, 1) or groups != 1 or pad not in (0, 1):
        return False
    N, _, H, W = x.shape
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    if Ho < 1 or Wo < 1 or H * W < 4:
        return False
    tiles = N * ((Ho + 1) // 2) * ((Wo + 1) // 2)
    if N >= 2 and tiles >= (WINO_MIN_TILES_BWD_BATCHED if backward else WINO_MIN_TILES_FWD_BATCHED):
        return True
    return tiles >= (WINO_MIN_TILES_BWD if backward else WINO_MIN_TILES_FWD)


def conv3x3_wgrad_eligible(x, weight, stride, padding, dilation, groups):
    if not (WINOGRAD_CONV and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return False
    one = lambda v, k: (v == k) if isinstance(v, int) else all(t == k for t in v)
    pad = padding if isinstance(padding, int) else (padding[0] if padding[0] == padding[1] else -1)
    if tuple(weight.shape[2:]) != (3, 3) or not one(stride, 1) or not one(dilation, 1) or groups != 1 or pad not in (0, 1):
        return False
    _, Ci, H, W = x.shape
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    # the kernel works on 64-pixel row segments: a mostly empty last segment (CAIN's 160-wide maps: 83 % used) loses
    fill = Wo / (64.0 * ((Wo + 63) // 64)) if Wo > 0 else 0.0
    if Ho > 0 and Wo > 0 and _wgrad_wino(x.shape[0], Ci, weight.shape[0], Ho, Wo):
        return True
    return Ci >= WGRAD_MIN_CI and Ho * Wo >= WGRAD_MIN_PIXELS and fill >= 0.85


# Weight gradients on a side stream.  In the backward of a first-order support pass the data-gradient chain (layer L's gx
# feeds layer L-1) is the critical path; the weight gradients hang off it and are only consumed by the update after the whole
# backward.  While overlap is switched on (per thread, by the caller that owns the autograd.grad() call) every fused conv
# op remembers the side stream at FORWARD time (its backward runs on autograd's device thread, where a thread-local flag
# would not be visible) and queues its weight gradient there; the caller joins before it touches the gradients.
_WG = threading.local()


_WG_SIDE = {}              # compute stream handle -> its side stream (persistent: streams and the library handles bound
_WG_LOCK = threading.Lock()    # to them are not created per pass or per worker thread)


def set_weight_gradient_overlap(on):
    _WG.stream = None
    if on:
        key = (torch.cuda.current_device(), _hip.current_stream())
        with _WG_LOCK:
            side = _WG_SIDE.get(key)
            if side is None:
                side = _WG_SIDE[key] = torch.cuda.Stream()
        _WG.stream = side
    _WG.on = bool(on)
    _WG.uses = {}          # id(weight) -> [number of fused conv ops that read it in this pass]


def _weight_use_counter(w):
    """A weight read by more than one op of the pass (the unfused support triplets, a plugin that shares layers) gets
    its contributions ADDED by autograd on the compute stream as soon as the second one is returned - before the caller's
    join.  Such weights keep their gradient on the compute stream (decided in backward, when the count is final)."""
    holder = _WG.uses.setdefault(id(w), [0])
    holder[0] += 1
    return holder


def weight_gradient_stream():
    return getattr(_WG, 'stream', None) if getattr(_WG, 'on', False) else None


def join_weight_gradients():
    """Make the current stream wait for every weight gradient queued on its side stream."""
    if not _WG_SIDE:
        return
    with _WG_LOCK:
        side = _WG_SIDE.get((torch.cuda.current_device(), _hip.current_stream()))
    if side is not None:
        torch.cuda.current_stream().wait_stream(side)


class _ConvBiasAct(torch.autograd.Function):
    """y = act(conv2d(x, w) + b).  Large 3x3 convolutions run on the savfi Winograd/MFMA kernel with bias and
    activation in its epilogue; the others stay on MIOpen with the bias add and activation as ONE in-place kernel
    on the output.  The backward computes act' * gy and the bias gradient in one pass, then the data gradient
    (savfi kernel or MIOpen) and the weight gradient (MIOpen).  First-order only (the backward is not
    differentiable): callers use the unfused ops under --second_order."""

    @staticmethod
    def forward(ctx, x, w, b, stride, padding, dilation, groups, slope, direct=False, cache=None, reflect=False, in_slope=None, defer=False):
        # conv -> ReLU -> conv chains (model_utils.MetaSequential): `defer` -- the consumer of y applies THIS layer's activation
        # derivative (the gradient arriving here is already d/dz); `in_slope` -- x is the activated output of a layer that deferred
        # its derivative to this one: it is folded into this layer's data gradient (kernel epilogue where there is one)
        pad = padding if isinstance(padding, int) else padding[0]
        ctx.in_slope, ctx.defer = in_slope, bool(defer)
        ctx.u_bwd, ctx.route, ctx.reflect = None, None, bool(reflect)
        assert not reflect or convk_eligible(x, w, stride, padding, dilation, groups, direct), "mirrored borders: direct kernel only"
        if convk_eligible(x, w, stride, padding, dilation, groups, direct):
            K = int(w.shape[-1])
            u_fwd, ctx.u_bwd = _filters('convk', w, True, bool(ctx.needs_input_grad[0]), cache)
            z = convk_tasks_pre(x, u_fwd, 1, w.shape[1], w.shape[0], K, b, 0, slope, pad, direct, reflect)
            ctx.route = 'convk'
        elif conv3x3_eligible(x, w, stride, padding, dilation, groups):
            want_bwd = ctx.needs_input_grad[0] and conv3x3_eligible(x, w, stride, padding, dilation, groups, backward=True)
            ctx.route = 'wino2' if wino_form2(x, w, pad) else 'wino'
            u_fwd, ctx.u_bwd = _filters(ctx.route, w, True, want_bwd, cache)
            z = conv3x3_tasks_pre(x, u_fwd, 1, w.shape[1], w.shape[0], b, 0, slope, pad, f2=ctx.route == 'wino2')
        else:
            z = torch.nn.functional.conv2d(x, w, None, stride, padding, dilation, groups)
            if not z.is_contiguous():
                z = z.contiguous()
            if b is not None or slope != 1.0:
                zb = b if b is not None else torch.zeros(z.shape[1], dtype=z.dtype, device=z.device)
                _hip.require_cuda(z, zb)
                N, C, H, W = z.shape
                lib = _hip.lib()
                _hip.launch("bias_act_fwd", lambda: _hip.check(lib.savfi_bias_act_fwd_f32(
                    z.data_ptr(), zb.data_ptr(), N, C, H * W, slope, _hip.current_stream()), "savfi_bias_act_fwd_f32"))
        ctx.conf = (stride, padding, dilation, groups, slope)
        ctx.direct, ctx.cache, ctx.w_version, ctx.has_bias = direct, cache, w._version, b is not None
        ctx.wg_stream = weight_gradient_stream() if x.is_cuda else None
        ctx.wg_uses = _weight_use_counter(w) if ctx.wg_stream is not None else None
        ctx.save_for_backward(x, w, z)
        return z

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        stride, padding, dilation, groups, slope = ctx.conf
        gy = gy.contiguous()
        N, C, H, W = y.shape
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2] and ctx.has_bias
        identity = slope == 1.0 or ctx.defer     # no activation, or its derivative already applied by the consumer: gz is gy itself,
        gz = gy if identity else torch.empty_like(gy)        # only the bias gradient is computed
        gb = torch.empty(C, dtype=gy.dtype, device=gy.device) if need_b else None
        mask, mslope = (x, ctx.in_slope) if (ctx.in_slope is not None and need_x) else (None, 1.0)
        pad = padding if isinstance(padding, int) else padding[0]
        K = int(w.shape[-1])
        # the bias gradient rides on the all-taps weight-gradient kernel's staging of the cotangent where this function has nothing else
        # to do with it (no activation, or its derivative left to the consumer): one pass over the map and two launches less per layer
        wgrad_is_convk = bool(need_w and (ctx.reflect or (_convk_geometry(w, stride, padding, dilation, groups) is not None
                                                          and (ctx.route == 'convk' or K == 3)
                                                          and convk_wgrad_preferred(K, w.shape[1], w.shape[0], H, W, ctx.direct, N))))
        fuse_b = bool(need_b and identity and wgrad_is_convk and convk_wgrad_tasks_sums_bias(x.shape, C, 1, K, pad, ctx.direct))
        if fuse_b:
            gb = None
        if (need_b and not fuse_b) or not identity:
            lib = _hip.lib()
            scratch = (torch.empty(_workspace_floats("savfi_bias_act_scratch_floats", N, C, H * W), dtype=gy.dtype, device=gy.device)
                       if gb is not None else None)
            _hip.launch("bias_act_bwd", lambda: _hip.check(lib.savfi_bias_act_bwd_f32(
                gy.data_ptr(), (gy if identity else y).data_ptr(), None if identity else gz.data_ptr(),
                None if gb is None else gb.data_ptr(), None if scratch is None else scratch.data_ptr(),
                N, C, H * W, 1.0 if identity else slope, _hip.current_stream()), "savfi_bias_act_bwd_f32"))
        gx = gw = None
        # the filter packed / transformed at forward time is only valid for the weight version the forward saw
        u_bwd = ctx.u_bwd if w._version == ctx.w_version else None
        ctx.u_bwd = None
        if need_x and ctx.route == 'convk':
            if u_bwd is None:
                u_bwd = _filters('convk', w, False, True, ctx.cache)[1]
            if ctx.reflect:     # gradient of the mirrored (padded) extent = the full data gradient of the unpadded convolution, folded
                gx = reflect_pad_bwd(convk_tasks_pre(gz, u_bwd, 1, w.shape[1], w.shape[0], K, None, 1, 1.0, 0, ctx.direct), pad)
            else:
                gx = convk_tasks_pre(gz, u_bwd, 1, w.shape[1], w.shape[0], K, None, 1, 1.0, pad, ctx.direct, mask=mask, mask_slope=mslope)
                mask = None
            need_x = False
        if need_w and ctx.reflect:
            res = convk_wgrad_tasks(x, gz, 1, K, pad, ctx.direct, True, want_bias=fuse_b)
            gw, gb = (res[0][0], res[1][0]) if fuse_b else (res[0], gb)
            need_w = False
        elif need_x and conv3x3_eligible(x, w, stride, padding, dilation, groups, backward=True):
            if u_bwd is not None and ctx.route in ('wino', 'wino2'):
                gx = conv3x3_tasks_pre(gz, u_bwd, 1, w.shape[1], w.shape[0], None, 1, 1.0, pad, mask=mask, mask_slope=mslope,
                                       f2=ctx.route == 'wino2')
                mask = None
            else:
                gx = conv3x3(gz, w, None, 1, 1.0, pad)
            need_x = False
        pair = lambda v: [v, v] if isinstance(v, int) else list(v)
        # 5x5 / 7x7 layers and plugins that asked for the direct form: weight gradient on the split-bf16 kernel as well
        if need_w and _convk_geometry(w, stride, padding, dilation, groups) is not None and \
                (ctx.route == 'convk' or K == 3) and convk_wgrad_preferred(K, w.shape[1], w.shape[0], gz.shape[2], gz.shape[3], ctx.direct, gz.shape[0]):
            res = convk_wgrad_tasks(x, gz, 1, K, pad, ctx.direct, want_bias=fuse_b)
            gw, gb = (res[0][0], res[1][0]) if fuse_b else (res[0], gb)
            need_w = False
        side = ctx.wg_stream if (ctx.wg_stream is not None and ctx.wg_uses[0] == 1) else None
        if need_w and side is not None:
            # gz was produced on this stream just above: the side stream picks up from here
            ready = torch.cuda.Event()
            ready.record()
            side.wait_event(ready)
            if conv3x3_wgrad_eligible(x, w, stride, padding, dilation, groups):
                gw = conv3x3_wgrad(x, gz, pad, stream=side.cuda_stream, extra_stream=side)
            else:
                with torch.cuda.stream(side):
                    _, gw, _ = torch.ops.aten.convolution_backward(gz, x, w, None, pair(stride), pair(padding), pair(dilation),
                                                                   False, [0, 0], groups, [False, True, False])
                gw.record_stream(torch.cuda.current_stream())      # consumed on this stream after the caller's join
            x.record_stream(side)
            gz.record_stream(side)
            need_w = False
        if need_w and conv3x3_wgrad_eligible(x, w, stride, padding, dilation, groups):
            gw = conv3x3_wgrad(x, gz, pad)
            need_w = False
        if need_x or need_w:
            gx2, gw2, _ = torch.ops.aten.convolution_backward(gz, x, w, None, pair(stride), pair(padding), pair(dilation),
                                                              False, [0, 0], groups, [need_x, need_w, False])
            gx = gx2 if need_x else gx
            gw = gw2 if need_w else gw
        if mask is not None and gx is not None:      # a route without the fused epilogue: the deferred derivative as its own pass
            gx = mask_by_activation(gx, mask, mslope)
        return gx, gw, gb, None, None, None, None, None, None, None, None, None, None


# --------------------------------------------------------------------------------------------
# The same fused conv for T tasks adapted in LOCKSTEP (reference: the sequential task loop meta_learning_system.py:366).
# Activations [n*T, C, H, W] are ordered sample-major (sample s = j*T + t belongs to task t), fast weights are stacked
# [T, Co, Ci, kh, kw] / [T, Co]: one launch per layer for the whole meta-batch.  Large-enough 3x3 layers run on the savfi
# kernels with a task index on the grid (per-task filter sets); everything else is ONE grouped MIOpen convolution --
# [n*T, C, H, W] viewed as [n, T*C, H, W] with groups = T, which the sample-major order makes a free view in and out.
# --------------------------------------------------------------------------------------------
# Thresholds from tools/tasks_bench.py (profiles/r02_tasks_bench_*.jsonl; T = 4 tasks x 2 samples).  Forward / data gradient:
# the savfi kernel wins or ties everywhere down to 24x32 maps (against 4 MIOpen calls: 1.1-2x); ONE grouped MIOpen call is
# ~15 % ahead on the 24x32 / 12x16 layers only (2 % of a step), not worth a second code path.  Weight gradient: the savfi
# kernel works on 64-pixel row segments and loses on maps under ~3000 px (512->512 @12x16: 311 us vs 111 us grouped);
# MIOpen's grouped weight gradient collapses on LARGE maps (10.5 ms at 384x512), so those always stay here, whatever Ci.
TASKS_MIN_TILES_FWD = 1          # N * ceil(Ho/2) * ceil(Wo/2) tiles over all tasks
TASKS_MIN_TILES_BWD = 1
TASKS_WGRAD_MIN_PIXELS = 3000


def _is3x3s1(weight, stride, padding, dilation):
    one = lambda v, k: (v == k) if isinstance(v, int) else all(t == k for t in v)
    pad = padding if isinstance(padding, int) else (padding[0] if padding[0] == padding[1] else -1)
    return tuple(weight.shape[-2:]) == (3, 3) and one(stride, 1) and one(dilation, 1) and pad in (0, 1)


def conv3x3_tasks_eligible(x, weight, stride, padding, dilation, backward=False):
    if not (WINOGRAD_CONV and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and _is3x3s1(weight, stride, padding, dilation)):
        return False
    pad = padding if isinstance(padding, int) else padding[0]
    N, _, H, W = x.shape
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    if Ho < 1 or Wo < 1 or H * W < 4:
        return False
    return N * ((Ho + 1) // 2) * ((Wo + 1) // 2) >= (TASKS_MIN_TILES_BWD if backward else TASKS_MIN_TILES_FWD)


def conv3x3_wgrad_tasks_eligible(x, weight, stride, padding, dilation):
    if not (WINOGRAD_CONV and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and _is3x3s1(weight, stride, padding, dilation)):
        return False
    pad = padding if isinstance(padding, int) else padding[0]
    N, Ci, H, W = x.shape
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    return Ho * Wo >= TASKS_WGRAD_MIN_PIXELS or (Ho > 0 and Wo > 0 and _wgrad_wino(N, Ci, weight.shape[1], Ho, Wo))


def conv3x3_tasks(x, weight, bias=None, mode=0, slope=1.0, pad=1, f2=False):
    """savfi_conv3x3_tasks_f32 without autograd: weight [T,Co,Ci,3,3], bias [T,Co] or None; sample n uses task n % T.  f2: the F(2x2) form."""
    fbit = 2 if f2 else 0
    x, weight = x.contiguous(), weight.contiguous()
    _hip.require_cuda(x, weight)
    N, _, H, W = x.shape
    T, Co, Ci = weight.shape[:3]
    assert N % T == 0 and tuple(weight.shape[3:]) == (3, 3) and x.shape[1] == (Ci if mode == 0 else Co), (x.shape, weight.shape, mode)
    I = Co if mode == 0 else Ci
    grow = 2 * (pad if mode == 0 else 2 - pad) - 2
    lib = _hip.lib()
    ws = torch.empty(_workspace_floats("savfi_conv3x3_tasks_workspace_floats", N, T, Ci, Co, H, W, int(pad), mode | fbit), dtype=x.dtype, device=x.device)
    out = torch.empty((N, I, H + grow, W + grow), dtype=x.dtype, device=x.device)
    _hip.launch(_conv3x3_name(Ci, Co, "fwd" if mode == 0 else "bwd_data"), lambda: _hip.check(lib.savfi_conv3x3_tasks_f32(
        x.data_ptr(), weight.data_ptr(), None if bias is None else bias.data_ptr(), out.data_ptr(), ws.data_ptr(),
        N, T, Ci, Co, H, W, int(pad), mode | fbit, float(slope), _hip.current_stream()), "savfi_conv3x3_tasks_f32"),
        flops=18.0 * Ci * Co * out.shape[2] * out.shape[3] * N if mode == 0 else 18.0 * Ci * Co * H * W * N)
    return out


# --------------------------------------------------------------------------------------------
# Filters of every layer in ONE launch per inner step.
#
# The fused kernels read a fast weight through a packed (convk) or Winograd-transformed (wino) copy, made per layer and pass: 270
# launches of ~9 us in a SepConv meta-iteration, 760 in a CAIN one (profiles/r03_*_one_iteration.txt).  The fast weights of a step
# are born together in mt_update: the first step records which of its outputs were packed, and how (the "plan" of that list of
# shapes); from then on every update packs those outputs straight away with savfi_*_filters_multi_f32, and convk_filters /
# conv3x3_filters find the result.  An entry keeps its weight tensor alive, so a data pointer cannot come back as another tensor
# while the entry exists; the weight's version is part of the match.
# --------------------------------------------------------------------------------------------
PREPACK = True
_pack_plans = {}        # tuple of the update's weight shapes -> {index: [kind, fwd, bwd]}
_last_update = None     # (signature, {data_ptr: index}, outputs) of the newest update
_prepacked = {}         # (kind, data_ptr) -> (weight, version, filters_fwd, filters_bwd)


def _filter_shape(weight):
    T, Co, Ci = (1,) + tuple(weight.shape[:2]) if weight.dim() == 4 else tuple(weight.shape[:3])
    return int(T), int(Co), int(Ci), int(weight.shape[-1])


def filters_after_update(outs):
    """Called with the fast weights an update just produced: pack / transform the ones the plan of this list names."""
    global _last_update
    _prepacked.clear()
    if not PREPACK or not outs or not outs[0].is_cuda:
        _last_update = None
        return
    sig = tuple(tuple(o.shape) for o in outs)
    _last_update = (sig, {o.data_ptr(): i for i, o in enumerate(outs)}, outs)
    plan = _pack_plans.get(sig)
    if not plan:
        return
    for kind in ('convk', 'wino', 'wino2'):
        jobs = [(outs[i], e[1], e[2]) for i, e in sorted(plan.items()) if e[0] == kind]
        for (w, _, _), (tf, tb) in zip(jobs, _filters_multi(kind, jobs)):
            _prepacked[(kind, w.data_ptr())] = (w, w._version, tf, tb)


def _filters_multi(kind, jobs):
    """[(weight, want_fwd, want_bwd)] -> [(filters_fwd, filters_bwd)] with ONE launch per 56 (layer, mode) jobs; the results are
    slices of one buffer."""
    if not jobs:
        return []
    lib = _hip.lib()
    dev = jobs[0][0].device
    sizes, total = [], 0
    for w, f, b in jobs:
        T, Co, Ci, K = _filter_shape(w)
        if kind == 'convk':
            nf = _workspace_floats("savfi_convk_filter_floats", T, Ci, Co, K, 0) if f else 0
            nb = _workspace_floats("savfi_convk_filter_floats", T, Ci, Co, K, 1) if b else 0
        else:
            fbit = 2 if kind == 'wino2' else 0
            nf = _workspace_floats("savfi_conv3x3_filter_floats", T, Ci, Co, 0 | fbit) if f else 0
            nb = _workspace_floats("savfi_conv3x3_filter_floats", T, Ci, Co, 1 | fbit) if b else 0
        nf, nb = (nf + 63) // 64 * 64, (nb + 63) // 64 * 64          # 256-byte aligned slices
        sizes.append((total, nf, total + nf, nb))
        total += nf + nb
    flat = torch.empty(total, dtype=torch.float32, device=dev)
    n = len(jobs)
    PA, IA = ctypes.c_void_p * n, ctypes.c_int * n
    pw, pf, pb = PA(), PA(), PA()
    aT, aCi, aCo, aK = IA(), IA(), IA(), IA()
    made = []
    for k, ((w, f, b), (of, nf, ob, nb)) in enumerate(zip(jobs, sizes)):
        T, Co, Ci, K = _filter_shape(w)
        tf = flat[of:of + nf] if nf else None
        tb = flat[ob:ob + nb] if nb else None
        pw[k], pf[k], pb[k] = w.data_ptr(), (tf.data_ptr() if nf else None), (tb.data_ptr() if nb else None)
        aT[k], aCi[k], aCo[k], aK[k] = T, Ci, Co, K
        made.append((tf, tb))
    if kind == 'convk':
        _hip.launch("convk_filters_multi", lambda: _hip.check(lib.savfi_convk_filters_multi_f32(
            pw, pf, pb, aT, aCi, aCo, aK, n, _hip.current_stream()), "savfi_convk_filters_multi_f32"))
    else:
        aF = IA(*([2 if kind == 'wino2' else 0] * n))
        _hip.launch("conv3x3_filters_multi", lambda: _hip.check(lib.savfi_conv3x3_filters_multi_form_f32(
            pw, pf, pb, aT, aCi, aCo, aF, n, _hip.current_stream()), "savfi_conv3x3_filters_multi_form_f32"))
    return made


def refresh_module_filters(modules):
    """After an in-place update of the modules' OWN weights (the outer optimizer step): re-make, in one launch per kind, the
    filters each module's cache holds for the previous version of its weight (CAIN: 500 single-layer packs per meta-iteration)."""
    if not PREPACK or torch.cuda.is_current_stream_capturing():
        return
    st = _hip.current_stream()
    jobs = {'convk': [], 'wino': [], 'wino2': []}
    for m in modules:
        cache, w = getattr(m, '_filters', None), getattr(m, 'weight', None)
        if not cache or w is None or not w.is_cuda or not w.is_contiguous():
            continue
        for key in reversed(list(cache)):
            kind, ptr, ver, shape, dev, stream = key
            if ptr == w.data_ptr() and shape == tuple(w.shape) and stream == st:
                if ver != w._version:
                    old = cache[key]
                    jobs[kind].append((m, key, (w.detach(), old[0] is not None, old[1] is not None)))
                break
    for kind, items in jobs.items():
        for (m, key, (w, _, _)), made in zip(items, _filters_multi(kind, [it[2] for it in items])):
            m._filters.pop(key, None)
            m._filters[(kind, w.data_ptr(), w._version, tuple(w.shape), w.device.index, st)] = made


# Memory-layout tags.  Two tensors of the SepConv tail keep the shape [4N, 51, H, W] while their MEMORY is unit-major, [H][W / 16][51][16]
# per sample: the taps between the last Subnet convolution and FunctionSepconvPair, and their cotangent on the way back.
# The layout travels as an attribute of the tensor object; producer and consumer name what they write / expect, and a tensor that reaches
# a consumer with another tag -- or a unit-major one that reaches a consumer which knows nothing of tags after a hook, a clone or a sum
# replaced the object -- fails here instead of being read as scrambled numbers.
UNIT16 = "unit16"


def tag_layout(t, layout):
    t._savfi_layout = layout
    return t


def layout_of(t):
    return getattr(t, "_savfi_layout", None)


def require_layout(t, layout, what):
    got = layout_of(t)
    if got != layout:
        raise SavfiLayoutError("%s: memory layout %r expected, the tensor is tagged %r (a unit-major tensor keeps the shape [N,51,H,W] "
                               "while its memory is [N][H][W/16][51][16]; only its producer's partner may read it)" % (what, layout, got))


class SavfiLayoutError(RuntimeError):
    pass


# Long-lived constant weights (sepconv/model.py: the four sub-networks' own parameters stacked into one task-batched layer; rebuilt when a
# parameter changes): their packed / transformed filters are made once per tensor, whichever function asks.  Keyed on the data pointer of a
# tensor the registry keeps alive, so the pointer cannot be recycled while the entry exists.
_const_weights = {}
_const_weights_lock = threading.Lock()      # --task_streams: the per-task Python threads register / evict concurrently


_CONST_WEIGHTS_MAX = 64         # a model that goes away without unregistering leaves its entries behind: oldest out (16 per SepConv net and stream)


def register_const_weight(w):
    with _const_weights_lock:
        while len(_const_weights) >= _CONST_WEIGHTS_MAX:
            _const_weights.pop(next(iter(_const_weights)), None)
        _const_weights[w.data_ptr()] = [w, w._version, {}]
    return w


def unregister_const_weight(w):
    with _const_weights_lock:
        _const_weights.pop(w.data_ptr(), None)


def _const_filters(kind, weight, fwd, bwd, make):
    """(filters_fwd, filters_bwd) of a registered constant weight (made on first use by `make(fwd, bwd)`), or None."""
    e = _const_weights.get(weight.data_ptr()) if _const_weights else None
    if e is None or e[1] != weight._version or e[0].shape != weight.shape or torch.cuda.is_current_stream_capturing():
        return None
    have = e[2].setdefault((kind, _hip.current_stream()), [None, None])
    need_f, need_b = fwd and have[0] is None, bwd and have[1] is None
    if need_f or need_b:
        pf, pb = make(need_f, need_b)
        if need_f:
            have[0] = pf
        if need_b:
            have[1] = pb
    return (have[0] if fwd else None), (have[1] if bwd else None)


def _prepacked_filters(kind, weight, fwd, bwd):
    hit = _prepacked.get((kind, weight.data_ptr())) if _prepacked else None
    if hit is None or hit[1] != weight._version or hit[0].shape != weight.shape or (fwd and hit[2] is None) or (bwd and hit[3] is None):
        return None
    return (hit[2] if fwd else None), (hit[3] if bwd else None)


def _note_filter_use(kind, weight, fwd, bwd):
    """A layer made its own filters from `weight`: if that is an output of the newest update, the plan of that update learns it."""
    if _last_update is None:
        return
    sig, index, _ = _last_update
    i = index.get(weight.data_ptr())
    if i is None or tuple(weight.shape) != sig[i]:
        return
    entry = _pack_plans.setdefault(sig, {}).get(i)
    if entry is None:
        _pack_plans[sig][i] = [kind, bool(fwd), bool(bwd)]
    elif entry[0] == kind:
        entry[1], entry[2] = entry[1] or bool(fwd), entry[2] or bool(bwd)


def conv3x3_filters(weight, fwd=True, bwd=True, f2=False):
    """savfi_conv3x3_filters_form_f32: the Winograd transforms of weight [T,Co,Ci,3,3] (or [Co,Ci,3,3]) for the forward pass and / or
    the data gradient, in ONE launch.  Returns (u_fwd, u_bwd); an entry is None when not asked for.  f2: the F(2x2) form whatever the
    channel counts (kind 'wino2': wino_form2) -- such filters go with conv3x3_tasks_pre(..., f2=True) only."""
    kind, fbit = ('wino2', 2) if f2 else ('wino', 0)
    weight = weight.contiguous()
    _hip.require_cuda(weight)
    T, Co, Ci = (1,) + tuple(weight.shape[:2]) if weight.dim() == 4 else tuple(weight.shape[:3])
    assert tuple(weight.shape[-2:]) == (3, 3) and (fwd or bwd), weight.shape
    ready = _prepacked_filters(kind, weight, fwd, bwd)
    if ready is not None:
        return ready

    def make(fwd, bwd):
        lib = _hip.lib()
        us = [torch.empty(_workspace_floats("savfi_conv3x3_filter_floats", T, Ci, Co, mode | fbit), dtype=weight.dtype, device=weight.device) if want else None
              for mode, want in ((0, fwd), (1, bwd))]
        _hip.launch("conv3x3_filters", lambda: _hip.check(lib.savfi_conv3x3_filters_form_f32(
            weight.data_ptr(), None if us[0] is None else us[0].data_ptr(), None if us[1] is None else us[1].data_ptr(), T, Ci, Co, fbit,
            _hip.current_stream()), "savfi_conv3x3_filters_form_f32"))
        return us[0], us[1]
    ready = _const_filters(kind, weight, fwd, bwd, make)
    if ready is not None:
        return ready
    _note_filter_use(kind, weight, fwd, bwd)
    return make(fwd, bwd)


def conv3x3_unit16_supported(x, w, pad):
    """Can the 3x3 layer (x [N,Ci,H,W], task weights w [T,Co,Ci,3,3], zero padding `pad`) write its result unit-major
    (savfi_conv3x3_tasks_pre_unit16_f32)?  It runs on the Winograd kernel, its width is a multiple of 16, no reduction split."""
    if not (x.is_cuda and w.dim() == 5 and conv3x3_tasks_eligible(x, w, 1, pad, 1) and not convk_eligible(x, w, 1, pad, 1, 1, False)):
        return False
    if wino_form2(x, w, pad):          # a small launch runs the F(2x2) form (kind 'wino2'), whose unit-major entry points are the form-0 ones
        return False
    N, Ci, H, W = x.shape
    return int(_hip.lib().savfi_conv3x3_unit16_supported(N, w.shape[0], Ci, w.shape[1], H, W, int(pad))) == 1


def conv3x3_in_unit16_supported(gy_shape, w, pad):
    """Can the data gradient of the 3x3 layer (cotangent of shape gy_shape, task weights w [T,Co,Ci,3,3]) read a unit-major cotangent
    (savfi_conv3x3_dgrad_in_unit16_f32)?"""
    N, Co, H, W = gy_shape
    return int(_hip.lib().savfi_conv3x3_in_unit16_supported(N, w.shape[0], w.shape[2], Co, H, W, int(pad))) == 1


def conv3x3_dgrad_in_unit16(gy, u, T, Ci, Co, pad):
    """savfi_conv3x3_dgrad_in_unit16_f32: the data gradient on a cotangent whose MEMORY is unit-major ([N][H][W/16][Co][16])."""
    gy = gy.contiguous()
    _hip.require_cuda(gy, u)
    N, _, H, W = gy.shape
    grow = 2 * (2 - pad) - 2
    out = torch.empty((N, Ci, H + grow, W + grow), dtype=gy.dtype, device=gy.device)
    lib = _hip.lib()
    _hip.launch(_conv3x3_name(Ci, Co, "bwd_data"), lambda: _hip.check(lib.savfi_conv3x3_dgrad_in_unit16_f32(
        gy.data_ptr(), u.data_ptr(), out.data_ptr(), N, T, Ci, Co, H, W, int(pad), _hip.current_stream()),
        "savfi_conv3x3_dgrad_in_unit16_f32"), flops=18.0 * Ci * Co * H * W * N)
    return out


def conv3x3_tasks_pre(x, u, T, Ci, Co, bias=None, mode=0, slope=1.0, pad=1, mask=None, mask_slope=1.0, out_unit16=False, f2=False):
    """savfi_conv3x3_tasks_pre_f32: conv3x3_tasks on a filter already transformed by conv3x3_filters (same mode).  `mask` (mode 1):
    the result is multiplied by (mask > 0 ? 1 : mask_slope) in the kernel's output stage (savfi_conv3x3_dgrad_masked_f32).
    `out_unit16` (mode 0): the result tensor has the usual shape [N,Co,Ho,Wo] but its MEMORY is unit-major, [N][Ho][Wo/16][Co][16]
    (savfi_conv3x3_tasks_pre_unit16_f32) -- for FunctionSepconvPair(..., taps_unit16=True) only.  `f2`: `u` is a kind-'wino2' filter
    (conv3x3_filters(..., f2=True)): the F(2x2) kernel (bit 1 of the C ABI's `mode`)."""
    x = x.contiguous()
    _hip.require_cuda(x, u)
    fbit = 2 if f2 else 0
    if mask is not None:
        assert mode == 1 and bias is None and slope == 1.0
        return _conv3x3_dgrad_masked(x, u, T, Ci, Co, pad, mask.contiguous(), mask_slope, fbit)
    N, _, H, W = x.shape
    assert N % T == 0 and x.shape[1] == (Ci if mode == 0 else Co), (x.shape, T, Ci, Co, mode)
    I = Co if mode == 0 else Ci
    grow = 2 * (pad if mode == 0 else 2 - pad) - 2
    lib = _hip.lib()
    if out_unit16:
        assert mode == 0 and not f2
        out = torch.empty((N, I, H + grow, W + grow), dtype=x.dtype, device=x.device)
        _hip.launch(_conv3x3_name(Ci, Co, "fwd"), lambda: _hip.check(lib.savfi_conv3x3_tasks_pre_unit16_f32(
            x.data_ptr(), u.data_ptr(), None if bias is None else bias.data_ptr(), out.data_ptr(), N, T, Ci, Co, H, W, int(pad), float(slope),
            _hip.current_stream()), "savfi_conv3x3_tasks_pre_unit16_f32"), flops=18.0 * Ci * Co * out.shape[2] * out.shape[3] * N)
        return out
    nws = _workspace_floats("savfi_conv3x3_tasks_pre_workspace_floats", N, T, Ci, Co, H, W, int(pad), mode | fbit)
    ws = torch.empty(nws, dtype=x.dtype, device=x.device) if nws else None
    out = torch.empty((N, I, H + grow, W + grow), dtype=x.dtype, device=x.device)
    name = ("conv3x3_" + ("fwd" if mode == 0 else "bwd_data")) if f2 else _conv3x3_name(Ci, Co, "fwd" if mode == 0 else "bwd_data")
    _hip.launch(name, lambda: _hip.check(lib.savfi_conv3x3_tasks_pre_f32(
        x.data_ptr(), u.data_ptr(), None if bias is None else bias.data_ptr(), out.data_ptr(), None if ws is None else ws.data_ptr(),
        N, T, Ci, Co, H, W, int(pad), mode | fbit, float(slope), _hip.current_stream()), "savfi_conv3x3_tasks_pre_f32"),
        flops=18.0 * Ci * Co * out.shape[2] * out.shape[3] * N if mode == 0 else 18.0 * Ci * Co * H * W * N)
    return out


def _conv3x3_dgrad_masked(gy, u, T, Ci, Co, pad, mask, mask_slope, fbit=0):
    N, _, H, W = gy.shape
    grow = 2 * (2 - pad) - 2
    out = torch.empty((N, Ci, H + grow, W + grow), dtype=gy.dtype, device=gy.device)
    assert mask.shape == out.shape, (mask.shape, out.shape)
    lib = _hip.lib()
    nws = _workspace_floats("savfi_conv3x3_tasks_pre_workspace_floats", N, T, Ci, Co, H, W, int(pad), 1 | fbit)
    ws = torch.empty(nws, dtype=gy.dtype, device=gy.device) if nws else None
    _hip.launch("conv3x3_bwd_data" if fbit else _conv3x3_name(Ci, Co, "bwd_data"), lambda: _hip.check(lib.savfi_conv3x3_dgrad_masked_form_f32(
        gy.data_ptr(), u.data_ptr(), mask.data_ptr(), float(mask_slope), out.data_ptr(), None if ws is None else ws.data_ptr(),
        N, T, Ci, Co, H, W, int(pad), fbit, _hip.current_stream()), "savfi_conv3x3_dgrad_masked_form_f32"), flops=18.0 * Ci * Co * H * W * N)
    return out


def mask_by_activation(g, y, slope):
    """g * (y > 0 ? 1 : slope): the (leaky) ReLU derivative from the activated output, one element-wise pass (savfi_bias_act_bwd_f32
    without the bias sums) -- the unfused form of the masked data-gradient epilogues."""
    g, y = g.contiguous(), y.contiguous()
    _hip.require_cuda(g, y)
    out = torch.empty_like(g)
    N, C = g.shape[:2]
    hw = g.numel() // (N * C)
    lib = _hip.lib()
    _hip.launch("bias_act_bwd", lambda: _hip.check(lib.savfi_bias_act_bwd_f32(
        g.data_ptr(), y.data_ptr(), out.data_ptr(), None, None, N, C, hw, float(slope), _hip.current_stream()), "savfi_bias_act_bwd_f32"))
    return out


class _MaskGrad(torch.autograd.Function):
    """identity whose backward multiplies by the activation derivative taken from the (activated) tensor itself: the consumer side of
    a deferred activation derivative when the consumer is not one of the fused convolutions."""

    @staticmethod
    def forward(ctx, x, slope):
        ctx.slope = slope
        ctx.save_for_backward(x)
        return x.view_as(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, = ctx.saved_tensors
        return mask_by_activation(g, x, ctx.slope), None


def mask_grad(x, slope):
    return _MaskGrad.apply(x, float(slope))


def convk_filters(weight, fwd=True, bwd=True):
    """savfi_convk_filters_f32: weight [T,Co,Ci,K,K] (or [Co,Ci,K,K]) packed as bf16 triples in MFMA fragment order for the
    forward pass and / or the data gradient of the direct K x K convolution, in one call.  Returns (p_fwd, p_bwd)."""
    weight = weight.contiguous()
    _hip.require_cuda(weight)
    T, Co, Ci = (1,) + tuple(weight.shape[:2]) if weight.dim() == 4 else tuple(weight.shape[:3])
    K = int(weight.shape[-1])
    assert weight.shape[-2] == K and (fwd or bwd), weight.shape
    ready = _prepacked_filters('convk', weight, fwd, bwd)
    if ready is not None:
        return ready

    def make(fwd, bwd):
        lib = _hip.lib()
        ps = [torch.empty(_workspace_floats("savfi_convk_filter_floats", T, Ci, Co, K, mode), dtype=torch.float32, device=weight.device)
              if want else None for mode, want in ((0, fwd), (1, bwd))]
        _hip.launch("convk_filters", lambda: _hip.check(lib.savfi_convk_filters_f32(
            weight.data_ptr(), None if ps[0] is None else ps[0].data_ptr(), None if ps[1] is None else ps[1].data_ptr(), T, Ci, Co, K,
            _hip.current_stream()), "savfi_convk_filters_f32"))
        return ps[0], ps[1]
    ready = _const_filters('convk', weight, fwd, bwd, make)
    if ready is not None:
        return ready
    _note_filter_use('convk', weight, fwd, bwd)
    return make(fwd, bwd)


def convk_tasks_pre(x, packed, T, Ci, Co, K, bias=None, mode=0, slope=1.0, pad=1, precise=False, reflect=False, mask=None, mask_slope=1.0):
    """savfi_convk_tasks_pre_f32: direct K x K convolution (mode 0, + bias + activation) or its data gradient (mode 1) on a
    filter packed by convk_filters (same mode); sample n uses filter set n % T.  `mask` (mode 1): the result is multiplied by
    (mask > 0 ? 1 : mask_slope) in the kernel's epilogue (savfi_convk_dgrad_masked_f32)."""
    x = x.contiguous()
    _hip.require_cuda(x, packed)
    N, _, H, W = x.shape
    assert N % T == 0 and x.shape[1] == (Ci if mode == 0 else Co), (x.shape, T, Ci, Co, mode)
    I = Co if mode == 0 else Ci
    p_eff = pad if mode == 0 else K - 1 - pad
    out = torch.empty((N, I, H + 2 * p_eff - K + 1, W + 2 * p_eff - K + 1), dtype=x.dtype, device=x.device)
    lib = _hip.lib()
    if mask is not None and (K != 3 or precise):
        # the masked epilogue exists for the 3 x 3 kernels of the conv chains; the same multiplication as a separate pass elsewhere
        return mask_by_activation(convk_tasks_pre(x, packed, T, Ci, Co, K, bias, mode, slope, pad, precise, reflect), mask, mask_slope)
    if mask is not None:
        assert mode == 1 and bias is None and slope == 1.0 and not reflect
        mask = mask.contiguous()
        assert mask.shape == out.shape, (mask.shape, out.shape)
        _hip.launch("convk_bwd_data", lambda: _hip.check(lib.savfi_convk_dgrad_masked_f32(
            x.data_ptr(), packed.data_ptr(), mask.data_ptr(), float(mask_slope), out.data_ptr(), N, T, Ci, Co, H, W, K, int(pad),
            int(bool(precise)), _hip.current_stream()), "savfi_convk_dgrad_masked_f32"), flops=2.0 * K * K * Ci * Co * N * H * W)
        return out
    _hip.launch("convk_fwd" if mode == 0 else "convk_bwd_data", lambda: _hip.check(lib.savfi_convk_tasks_pre_reflect_f32(
        x.data_ptr(), packed.data_ptr(), None if bias is None else bias.data_ptr(), out.data_ptr(), N, T, Ci, Co, H, W, K, int(pad),
        mode, float(slope), int(bool(precise)), int(bool(reflect)), _hip.current_stream()), "savfi_convk_tasks_pre_reflect_f32"),
        flops=2.0 * K * K * Ci * Co * N * (out.shape[2] * out.shape[3] if mode == 0 else H * W))
    return out


def convk_wgrad_tasks_sums_bias(x_shape, co, T, K, pad, precise=False):
    """Will convk_wgrad_tasks(..., want_bias=True) hand out the bias gradient with the weight gradient (the all-taps 3 x 3 kernel does)?"""
    N, Ci, H, W = x_shape
    return bool(WGRAD_FUSE_BIAS and not precise and _hip.lib().savfi_convk_wgrad_sums_bias(N, T, Ci, co, H, W, K, int(pad)))


def convk_wgrad_tasks(x, gz, T, K, pad, precise=False, reflect=False, want_bias=False):
    """savfi_convk_wgrad_tasks_f32: gw [T,Co,Ci,K,K], gw[t] over the samples n % T == t (direct K x K form, split-bf16 MFMAs).
    want_bias (only where convk_wgrad_tasks_sums_bias says yes): returns (gw, gb [T,Co]) -- the sums of gz ride on the kernel's staging of it."""
    x, gz = x.contiguous(), gz.contiguous()
    _hip.require_cuda(x, gz)
    N, Ci, H, W = x.shape
    Co = gz.shape[1]
    assert N % T == 0 and tuple(gz.shape) == (N, Co, H + 2 * pad - K + 1, W + 2 * pad - K + 1), (x.shape, gz.shape, K, pad, T)
    lib = _hip.lib()
    ws = torch.empty(_workspace_floats("savfi_convk_wgrad_workspace_floats", N, T, Ci, Co, H, W, K, int(pad)), dtype=x.dtype, device=x.device)
    gw = torch.empty((T, Co, Ci, K, K), dtype=x.dtype, device=x.device)
    flops = 2.0 * K * K * Ci * Co * N * gz.shape[2] * gz.shape[3]
    if want_bias:
        assert not precise and convk_wgrad_tasks_sums_bias(x.shape, Co, T, K, pad), "ask convk_wgrad_tasks_sums_bias first"
        gb = torch.empty((T, Co), dtype=x.dtype, device=x.device)
        _hip.launch("convk_wgrad", lambda: _hip.check(lib.savfi_convk_wgrad_tasks_bias_f32(
            x.data_ptr(), gz.data_ptr(), gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), N, T, Ci, Co, H, W, K, int(pad),
            int(bool(reflect)), _hip.current_stream()), "savfi_convk_wgrad_tasks_bias_f32"), flops=flops)
        return gw, gb
    _hip.launch("convk_wgrad", lambda: _hip.check(lib.savfi_convk_wgrad_tasks_reflect_f32(
        x.data_ptr(), gz.data_ptr(), gw.data_ptr(), ws.data_ptr(), N, T, Ci, Co, H, W, K, int(pad), int(bool(precise)),
        int(bool(reflect)), _hip.current_stream()), "savfi_convk_wgrad_tasks_reflect_f32"), flops=flops)
    return gw


def reflect_pad_bwd(gp, pad, add=None):
    """savfi_reflect_pad_bwd_f32: the adjoint of nn.ReflectionPad2d(pad) as a gather; gp [N,C,H+2p,W+2p] -> [N,C,H,W].
    add [N,C,H,W]: a second cotangent of the unpadded map, added in the same pass (savfi_reflect_pad_bwd_add_f32)."""
    gp = gp.contiguous()
    _hip.require_cuda(gp)
    N, C, Hp, Wp = gp.shape
    H, W = Hp - 2 * pad, Wp - 2 * pad
    gx = torch.empty((N, C, H, W), dtype=gp.dtype, device=gp.device)
    lib = _hip.lib()
    if add is not None:
        add = add.contiguous()
        _hip.require_cuda(add)
        assert tuple(add.shape) == (N, C, H, W) and add.dtype == gp.dtype, (add.shape, gp.shape, pad)
        _hip.launch("reflect_pad_bwd", lambda: _hip.check(lib.savfi_reflect_pad_bwd_add_f32(
            gp.data_ptr(), add.data_ptr(), gx.data_ptr(), N * C, H, W, int(pad), _hip.current_stream()), "savfi_reflect_pad_bwd_add_f32"),
            nbytes=4 * (gp.numel() + 2 * gx.numel()))
        return gx
    _hip.launch("reflect_pad_bwd", lambda: _hip.check(lib.savfi_reflect_pad_bwd_f32(
        gp.data_ptr(), gx.data_ptr(), N * C, H, W, int(pad), _hip.current_stream()), "savfi_reflect_pad_bwd_f32"),
        nbytes=4 * (gp.numel() + gx.numel()))
    return gx


class _ReflectPad(torch.autograd.Function):
    """nn.ReflectionPad2d(pad): savfi_reflect_pad_fwd_f32, the adjoint as ONE gather launch (savfi_reflect_pad_bwd_f32) where ATen's
    reflection_pad2d_backward zero-fills its result and scatters with atomics (two launches, order-dependent sums) -- the layers
    whose maps are too small for the mirrored staging of the direct kernels (CAIN at 64x64: 252 pads per meta-iteration)."""

    @staticmethod
    def forward(ctx, x, pad):
        ctx.pad = int(pad)
        return _reflect_pad_fwd(x, ctx.pad)

    @staticmethod
    def backward(ctx, g):
        return reflect_pad_bwd(g, ctx.pad), None


def _reflect_pad_fwd(x, p):
    x = x.contiguous()
    _hip.require_cuda(x)
    N, C, H, W = x.shape
    assert 0 <= p < H and p < W, (x.shape, p)
    if N * C > 65535:          # beyond the launch grid's plane axis: ATen's kernel
        return torch.nn.functional.pad(x, (p,) * 4, mode='reflect')
    xp = torch.empty((N, C, H + 2 * p, W + 2 * p), dtype=x.dtype, device=x.device)
    lib = _hip.lib()
    _hip.launch("reflect_pad", lambda: _hip.check(lib.savfi_reflect_pad_fwd_f32(
        x.data_ptr(), xp.data_ptr(), N * C, H, W, p, _hip.current_stream()), "savfi_reflect_pad_fwd_f32"),
        nbytes=4 * (x.numel() + xp.numel()))
    return xp


class _ReflectPadSkip(torch.autograd.Function):
    """(nn.ReflectionPad2d(pad)(x), x): the padded map for a convolution and the map itself for a connection round it (CAIN's RCAB,
    reference model_utils.py:957-990).  The two cotangents of x arrive in ONE backward call and leave as one pass, fold(g_pad) + g_skip
    (savfi_reflect_pad_bwd_add_f32), where autograd would add the fold's result and the skip gradient in an element-wise kernel of its own."""

    @staticmethod
    def forward(ctx, x, pad):
        ctx.pad = int(pad)
        ctx.set_materialize_grads(False)
        return _reflect_pad_fwd(x, ctx.pad), x.view_as(x)

    @staticmethod
    def backward(ctx, g_pad, g_skip):
        if g_pad is None:
            return g_skip, None
        return reflect_pad_bwd(g_pad, ctx.pad, add=g_skip), None


def reflect_pad_with_skip(x, pad):
    """(nn.ReflectionPad2d(pad)(x), x) with the two gradients of x added inside the fold (_ReflectPadSkip); passes that need a
    differentiable backward, CPU tensors and other dtypes: ATen's pad and x itself."""
    if double_backward() or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4:
        return torch.nn.functional.pad(x, (int(pad),) * 4, mode='reflect'), x
    return _ReflectPadSkip.apply(x, int(pad))


def reflect_pad(x, pad):
    """nn.ReflectionPad2d(pad)(x).  First-order passes on CUDA tensors take the gather adjoint; --second_order (whose backward
    must itself be differentiable) and CPU tensors stay on ATen."""
    if double_backward() or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4:
        return torch.nn.functional.pad(x, (int(pad),) * 4, mode='reflect')
    return _ReflectPad.apply(x, int(pad))


def convk_reflect_eligible(x, weight, pad):
    """MetaConvNorm (ReflectionPad2d(pad) + K x K convolution, K = 2 pad + 1) as ONE direct convolution that mirrors the border
    while it stages its tile?  Where the layer would take the direct kernel anyway (forward, data gradient and weight gradient
    all live in csrc/convk*.hip there) and the fast weight is a plain [Co,Ci,K,K] tensor."""
    if weight.dim() != 4 or x.dim() != 4 or pad < 1 or int(weight.shape[-1]) != 2 * pad + 1:
        return False
    if pad >= x.shape[2] or pad >= x.shape[3]:
        return False
    return convk_eligible(x, weight, 1, pad, 1, 1, False)


WGRAD_FUSE_BIAS = True     # A/B: False = the bias sums as a pass of their own (rounds 2-4)


def conv3x3_wgrad_tasks_sums_bias(x_shape, co, T, pad):
    """Will conv3x3_wgrad_tasks(..., want_bias=True) hand out the bias gradient with the weight gradient (the Winograd form does)?"""
    N, Ci, H, W = x_shape
    return WGRAD_FUSE_BIAS and _wgrad_wino(N, Ci, co, H + 2 * pad - 2, W + 2 * pad - 2)


def conv3x3_wgrad_tasks(x, gz, T, pad=1, stream=None, extra_stream=None, want_bias=False):
    """savfi_conv3x3_wgrad_tasks_f32: gw [T,Co,Ci,3,3], gw[t] over the samples n % T == t.  `stream` / `extra_stream`: as in
    conv3x3_wgrad (launch on a side stream, buffers from the current stream's pool).  want_bias (only where
    conv3x3_wgrad_tasks_sums_bias says yes): returns (gw, gb [T,Co]) -- the sums of gz ride on the weight gradient's read of it."""
    x, gz = x.contiguous(), gz.contiguous()
    _hip.require_cuda(x, gz)
    N, Ci, H, W = x.shape
    Co = gz.shape[1]
    assert N % T == 0 and tuple(gz.shape) == (N, Co, H + 2 * pad - 2, W + 2 * pad - 2), (x.shape, gz.shape, pad, T)
    lib = _hip.lib()
    form = "wino_" if _wgrad_wino(N, Ci, Co, H + 2 * pad - 2, W + 2 * pad - 2) else ""
    gw = torch.empty((T, Co, Ci, 3, 3), dtype=x.dtype, device=x.device)
    if want_bias:
        assert form == "wino_" and WGRAD_FUSE_BIAS, "ask conv3x3_wgrad_tasks_sums_bias first"
        ws = torch.empty(_workspace_floats("savfi_conv3x3_wgrad_wino_tasks_bias_workspace_floats", N, T, Ci, Co, H, W, int(pad)), dtype=x.dtype, device=x.device)
        gb = torch.empty((T, Co), dtype=x.dtype, device=x.device)
        if extra_stream is not None:
            for t in (ws, gw, gb):
                t.record_stream(extra_stream)
        _hip.launch("conv3x3_wgrad", lambda: _hip.check(lib.savfi_conv3x3_wgrad_wino_tasks_bias_f32(
            x.data_ptr(), gz.data_ptr(), gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), N, T, Ci, Co, H, W, int(pad),
            _hip.current_stream() if stream is None else stream), "savfi_conv3x3_wgrad_wino_tasks_bias_f32"),
            flops=18.0 * Ci * Co * gz.shape[2] * gz.shape[3] * N)
        return gw, gb
    ws = torch.empty(_workspace_floats("savfi_conv3x3_wgrad_%stasks_workspace_floats" % form, N, T, Ci, Co, H, W, int(pad)), dtype=x.dtype, device=x.device)
    if extra_stream is not None:
        ws.record_stream(extra_stream)
        gw.record_stream(extra_stream)
    entry = "savfi_conv3x3_wgrad_%stasks_f32" % form
    _hip.launch("conv3x3_wgrad", lambda: _hip.check(getattr(lib, entry)(
        x.data_ptr(), gz.data_ptr(), gw.data_ptr(), ws.data_ptr(), N, T, Ci, Co, H, W, int(pad),
        _hip.current_stream() if stream is None else stream), entry), flops=18.0 * Ci * Co * gz.shape[2] * gz.shape[3] * N)
    return gw


TASKS_GROUPED_MAX_PIXELS = 3000      # MIOpen's grouped solvers: fine on small maps, 5-30x slower than T plain calls on large ones


def _grouped_ok(x):
    return x.shape[2] * x.shape[3] < TASKS_GROUPED_MAX_PIXELS


def conv2d_tasks(x, weight, bias, stride=1, padding=0, dilation=1):
    """conv2d of [n*T, Ci, H, W] (sample-major) with per-task weights [T, Co, Ci, kh, kw] / bias [T, Co], composed of
    differentiable torch ops (second order, bias-free layers, CPU host-logic tests): ONE grouped convolution on small maps,
    one plain convolution per task on large ones (tools/tasks_bench.py: MIOpen's grouped kernels take 1.5-10 ms there)."""
    T, Co, Ci = weight.shape[:3]
    N, _, H, W = x.shape
    n = N // T
    if _grouped_ok(x) or not x.is_cuda:
        z = torch.nn.functional.conv2d(x.reshape(n, T * Ci, H, W), weight.reshape(T * Co, Ci, *weight.shape[3:]),
                                       None if bias is None else bias.reshape(T * Co), stride, padding, dilation, T)
        return z.reshape(N, Co, z.shape[2], z.shape[3])
    xs = x.reshape(n, T, Ci, H, W)
    zs = [torch.nn.functional.conv2d(xs[:, t], weight[t], None if bias is None else bias[t], stride, padding, dilation)
          for t in range(T)]
    z = torch.stack(zs, 1)                                     # [n, T, Co, Ho, Wo]: sample-major again
    return z.reshape(N, Co, z.shape[3], z.shape[4])


class _ConvBiasActTasks(torch.autograd.Function):
    """y = act(conv2d(x[s], w[s % T]) + b[s % T]) for every sample s.  First-order only (like _ConvBiasAct)."""

    @staticmethod
    def forward(ctx, x, w, b, stride, padding, dilation, slope, direct=False, in_slope=None, defer=False, out_unit16=False):
        x = x.contiguous()
        ctx.in_slope, ctx.defer = in_slope, bool(defer)          # see _ConvBiasAct.forward
        # out_unit16 == 2: the cotangent that comes back is unit-major as well (FunctionSepconvPair(..., grads_unit16=True)): only the
        # data gradient may be asked for (the caller has checked conv3x3_in_unit16_supported and that w, b carry no gradient)
        ctx.gy_unit16 = int(out_unit16) == 2
        out_unit16 = bool(out_unit16)
        T, Co, Ci = w.shape[:3]
        N, _, H, W = x.shape
        n = N // T
        pad = padding if isinstance(padding, int) else padding[0]
        ctx.u_bwd, ctx.route = None, None
        if out_unit16:
            # the result's MEMORY is unit-major (its only consumer is FunctionSepconvPair(taps_unit16=True)); the cotangent that comes back
            # is laid out like the shape says, so nothing changes in backward.  The caller has asked conv3x3_unit16_supported.
            assert slope == 1.0 and not defer and conv3x3_unit16_supported(x, w, pad), "unit-major output: Winograd route, no activation"
        if convk_eligible(x, w, stride, padding, dilation, 1, direct):
            u_fwd, ctx.u_bwd = convk_filters(w, True, bool(ctx.needs_input_grad[0]))
            z = convk_tasks_pre(x, u_fwd, T, Ci, Co, int(w.shape[-1]), b, 0, slope, pad, direct)
            ctx.route = 'convk'
        elif conv3x3_tasks_eligible(x, w, stride, padding, dilation):
            # both filter transforms of this layer in one launch: the data gradient of the same step will want the other one
            want_bwd = ctx.needs_input_grad[0] and conv3x3_tasks_eligible(x, w, stride, padding, dilation, backward=True)
            f2 = wino_form2(x, w, pad)
            assert not (f2 and out_unit16), "unit-major output: not on the F(2x2) form of a small launch (conv3x3_unit16_supported says so)"
            u_fwd, ctx.u_bwd = conv3x3_filters(w, True, want_bwd, f2=f2)
            z = conv3x3_tasks_pre(x, u_fwd, T, Ci, Co, b, 0, slope, pad, out_unit16=out_unit16, f2=f2)
            ctx.route = 'wino2' if f2 else 'wino'
        else:
            if _grouped_ok(x):
                z = torch.nn.functional.conv2d(x.view(n, T * Ci, H, W), w.reshape(T * Co, Ci, *w.shape[3:]), None, stride, padding,
                                               dilation, T)
                if not z.is_contiguous():
                    z = z.contiguous()
            else:           # large map, no savfi kernel (5x5 / 7x7 / strided): one MIOpen call per task, results interleaved
                xs = x.view(n, T, Ci, H, W)
                z = torch.stack([torch.nn.functional.conv2d(xs[:, t], w[t], None, stride, padding, dilation) for t in range(T)], 1)
                z = z.view(N, Co, z.shape[3], z.shape[4])
            if b is not None or slope != 1.0:
                zb = b if b is not None else torch.zeros((T, Co), dtype=z.dtype, device=z.device)
                _hip.require_cuda(z, zb)
                lib = _hip.lib()
                hw = z.shape[-2] * z.shape[-1]
                _hip.launch("bias_act_fwd", lambda: _hip.check(lib.savfi_bias_act_fwd_f32(
                    z.data_ptr(), zb.data_ptr(), n, T * Co, hw, slope, _hip.current_stream()), "savfi_bias_act_fwd_f32"))
            z = z.view(N, Co, z.shape[-2], z.shape[-1])
        ctx.conf = (stride, padding, dilation, slope)
        ctx.has_bias, ctx.direct, ctx.w_version = b is not None, direct, w._version
        ctx.wg_stream = weight_gradient_stream() if x.is_cuda else None
        ctx.wg_uses = _weight_use_counter(w) if ctx.wg_stream is not None else None
        ctx.save_for_backward(x, w, z)
        if out_unit16:
            tag_layout(z, UNIT16)           # the tensor's shape does not say how its memory is laid out: its consumer checks the tag
        return z

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        stride, padding, dilation, slope = ctx.conf
        # the cotangent's memory layout is a contract between two autograd functions that its shape cannot carry: the producer tags the
        # tensor, and anything in between (a hook, an accumulation of two consumers' gradients, a clone) loses or contradicts the tag
        require_layout(gy, UNIT16 if ctx.gy_unit16 else None, "the cotangent of conv_bias_act_tasks(out_unit16=%d)" % (2 if ctx.gy_unit16 else 1))
        gy = gy.contiguous()
        T, Co, Ci = w.shape[:3]
        N, _, Ho, Wo = y.shape
        n = N // T
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        if ctx.gy_unit16:
            assert not need_w and not (need_b and ctx.has_bias) and slope == 1.0 and ctx.in_slope is None and ctx.route == 'wino', \
                "a unit-major cotangent: data gradient of the Winograd route only"
            if not need_x:
                return (None,) * 11
            pad = padding if isinstance(padding, int) else padding[0]
            u_bwd = ctx.u_bwd if w._version == ctx.w_version else None
            ctx.u_bwd = None
            if u_bwd is None:
                u_bwd = conv3x3_filters(w, False, True)[1]
            return (conv3x3_dgrad_in_unit16(gy, u_bwd, T, Ci, Co, pad),) + (None,) * 10
        identity = slope == 1.0 or ctx.defer
        gz = gy if identity else torch.empty_like(gy)
        need_b = need_b and ctx.has_bias
        mask, mslope = (x, ctx.in_slope) if (ctx.in_slope is not None and need_x) else (None, 1.0)
        # the bias gradient rides on the weight gradient's read of gz where that is the Winograd form or the all-taps direct kernel and this
        # function has nothing else to do with the cotangent (no activation derivative to apply here): one pass over the map less
        pad_ = padding if isinstance(padding, int) else padding[0]
        K_ = int(w.shape[-1])
        wgrad_is_convk = bool(need_w and _convk_geometry(w, stride, padding, dilation, 1) is not None and (ctx.route == 'convk' or K_ == 3)
                              and convk_wgrad_preferred(K_, Ci, Co, Ho, Wo, ctx.direct, x.shape[0]))
        wgrad_is_wino3 = need_w and not wgrad_is_convk and conv3x3_wgrad_tasks_eligible(x, w, stride, padding, dilation)
        # (not beside the Winograd weight gradient on the SIDE stream: a bias gradient autograd consumes on the compute stream -- stacked
        # biases, a bias shared between ops -- must be produced there; the side-stream guard counts uses of the WEIGHT only)
        wino3_on_side = wgrad_is_wino3 and ctx.wg_stream is not None and ctx.wg_uses[0] == 1
        fuse_b = bool(need_b and identity and ((wgrad_is_wino3 and not wino3_on_side and conv3x3_wgrad_tasks_sums_bias(x.shape, Co, T, pad_))
                                               or (wgrad_is_convk and convk_wgrad_tasks_sums_bias(x.shape, Co, T, K_, pad_, ctx.direct))))
        gb = torch.empty((T, Co), dtype=gy.dtype, device=gy.device) if (need_b and not fuse_b) else None
        if (need_b and not fuse_b) or not identity:
            lib = _hip.lib()
            scratch = (torch.empty(_workspace_floats("savfi_bias_act_scratch_floats", n, T * Co, Ho * Wo), dtype=gy.dtype, device=gy.device)
                       if gb is not None else None)
            _hip.launch("bias_act_bwd", lambda: _hip.check(lib.savfi_bias_act_bwd_f32(
                gy.data_ptr(), (gy if identity else y).data_ptr(), None if identity else gz.data_ptr(),
                None if gb is None else gb.data_ptr(), None if scratch is None else scratch.data_ptr(),
                n, T * Co, Ho * Wo, 1.0 if identity else slope, _hip.current_stream()), "savfi_bias_act_bwd_f32"))
        gx = gw = None
        pad = padding if isinstance(padding, int) else padding[0]
        K = int(w.shape[-1])
        u_bwd = ctx.u_bwd if w._version == ctx.w_version else None      # valid for the weight version the forward saw only
        ctx.u_bwd = None
        if need_x and ctx.route == 'convk':
            if u_bwd is None:
                u_bwd = convk_filters(w, False, True)[1]
            gx = convk_tasks_pre(gz, u_bwd, T, Ci, Co, K, None, 1, 1.0, pad, ctx.direct, mask=mask, mask_slope=mslope)
            mask = None
            need_x = False
        elif need_x and conv3x3_tasks_eligible(x, w, stride, padding, dilation, backward=True):
            if u_bwd is not None and ctx.route in ('wino', 'wino2'):
                gx = conv3x3_tasks_pre(gz, u_bwd, T, Ci, Co, None, 1, 1.0, pad, mask=mask, mask_slope=mslope, f2=ctx.route == 'wino2')
                mask = None
            else:
                gx = conv3x3_tasks(gz, w, None, 1, 1.0, pad, f2=ctx.route == 'wino2')
            need_x = False
        if wgrad_is_convk:
            gw = convk_wgrad_tasks(x, gz, T, K, pad, ctx.direct, want_bias=fuse_b)
            if fuse_b:
                gw, gb = gw
            need_w = False
        if need_w and conv3x3_wgrad_tasks_eligible(x, w, stride, padding, dilation):
            side = ctx.wg_stream if (ctx.wg_stream is not None and ctx.wg_uses[0] == 1) else None
            if side is not None:     # beside the data-gradient chain (see _ConvBiasAct.backward); joined by the caller
                ready = torch.cuda.Event()
                ready.record()
                side.wait_event(ready)
                gw = conv3x3_wgrad_tasks(x, gz, T, pad, stream=side.cuda_stream, extra_stream=side, want_bias=fuse_b)
                x.record_stream(side)
                gz.record_stream(side)
            else:
                gw = conv3x3_wgrad_tasks(x, gz, T, pad, want_bias=fuse_b)
            if fuse_b:
                gw, gb = gw
            need_w = False
        if need_x or need_w:
            pair = lambda v: [v, v] if isinstance(v, int) else list(v)
            H, W = x.shape[2:]
            if _grouped_ok(x):
                gx2, gw2, _ = torch.ops.aten.convolution_backward(
                    gz.view(n, T * Co, Ho, Wo), x.view(n, T * Ci, H, W), w.reshape(T * Co, Ci, *w.shape[3:]), None, pair(stride),
                    pair(padding), pair(dilation), False, [0, 0], T, [need_x, need_w, False])
                if need_x:
                    gx = gx2.contiguous().view(N, Ci, H, W)
                if need_w:
                    gw = gw2.view(w.shape)
            else:
                gzs, xs = gz.view(n, T, Co, Ho, Wo), x.view(n, T, Ci, H, W)
                per = [torch.ops.aten.convolution_backward(gzs[:, t].contiguous(), xs[:, t].contiguous(), w[t], None, pair(stride),
                                                           pair(padding), pair(dilation), False, [0, 0], 1, [need_x, need_w, False])
                       for t in range(T)]
                if need_x:
                    gx = torch.stack([p[0] for p in per], 1).view(N, Ci, H, W)
                if need_w:
                    gw = torch.stack([p[1] for p in per], 0)
        if mask is not None and gx is not None:
            gx = mask_by_activation(gx, mask, mslope)
        return gx, gw, gb, None, None, None, None, None, None, None, None


def conv_bias_act_tasks(x, weight, bias, stride=1, padding=0, dilation=1, slope=0.0, direct=False, in_slope=None, defer=False,
                        out_unit16=False):
    """act(conv2d(x[s], weight[s % T]) + bias[s % T]): the lockstep form of conv_bias_act (`in_slope`, `defer`: conv_bias_act).
    `out_unit16`: the result's memory is unit-major (conv3x3_tasks_pre; only after conv3x3_unit16_supported said yes, slope 1)."""
    z = _ConvBiasActTasks.apply(x, weight, bias, stride, padding, dilation, float(slope), bool(direct),
                                None if in_slope is None else float(in_slope), bool(defer), int(out_unit16))
    return tag_layout(z, UNIT16) if out_unit16 else z


@functools.lru_cache(maxsize=None)
def _workspace_floats(query, *shape):
    """Size queries of the C ABI are pure functions of the shape: ask once per shape."""
    n = int(getattr(_hip.lib(), query)(*shape))
    if n < 0:
        _hip.check(n, query)
    return n


def conv3x3(x, weight, bias=None, mode=0, slope=1.0, pad=1):
    """savfi_conv3x3_f32 without autograd.  mode 0: act(conv2d(x, weight, padding=pad) + bias); mode 1: the data
    gradient of that convolution (x = gy [N,Co,H,W] -> gx [N,Ci,H+2-2pad,W+2-2pad])."""
    x = x.contiguous()
    weight = weight.contiguous()
    _hip.require_cuda(x, weight)
    N, _, H, W = x.shape
    Co, Ci = weight.shape[:2]
    assert tuple(weight.shape[2:]) == (3, 3) and x.shape[1] == (Ci if mode == 0 else Co), (x.shape, weight.shape, mode)
    K, I = (Ci, Co) if mode == 0 else (Co, Ci)
    grow = 2 * (pad if mode == 0 else 2 - pad) - 2
    lib = _hip.lib()
    ws = torch.empty(_workspace_floats("savfi_conv3x3_workspace_floats", N, Ci, Co, H, W, int(pad), mode), dtype=x.dtype, device=x.device)
    out = torch.empty((N, I, H + grow, W + grow), dtype=x.dtype, device=x.device)
    _hip.launch(_conv3x3_name(Ci, Co, "fwd" if mode == 0 else "bwd_data"), lambda: _hip.check(lib.savfi_conv3x3_f32(
        x.data_ptr(), weight.data_ptr(), None if bias is None else bias.data_ptr(), out.data_ptr(), ws.data_ptr(),
        N, Ci, Co, H, W, int(pad), mode, float(slope), _hip.current_stream()), "savfi_conv3x3_f32"),
        flops=18.0 * Ci * Co * out.shape[2] * out.shape[3] * N if mode == 0 else 18.0 * Ci * Co * H * W * N)
    return out


def conv3x3_wgrad(x, gz, pad=1, stream=None, extra_stream=None):
    """savfi_conv3x3_wgrad_f32: weight gradient [Co,Ci,3,3] of conv2d(x, w, padding=pad) for the cotangent gz.
    `stream` (raw handle) launches on another stream than torch's current one; the result and the workspace are still
    allocated from the current stream's pool (safe: that stream has been made to wait for this one) and registered with
    `extra_stream` so that they are not recycled while it still uses them."""
    x, gz = x.contiguous(), gz.contiguous()
    _hip.require_cuda(x, gz)
    N, Ci, H, W = x.shape
    Co = gz.shape[1]
    assert tuple(gz.shape) == (N, Co, H + 2 * pad - 2, W + 2 * pad - 2), (x.shape, gz.shape, pad)
    if _wgrad_wino(N, Ci, Co, H + 2 * pad - 2, W + 2 * pad - 2):
        return conv3x3_wgrad_tasks(x, gz, 1, pad, stream, extra_stream)[0]
    lib = _hip.lib()
    ws = torch.empty(_workspace_floats("savfi_conv3x3_wgrad_workspace_floats", N, Ci, Co, H, W, int(pad)), dtype=x.dtype, device=x.device)
    gw = torch.empty((Co, Ci, 3, 3), dtype=x.dtype, device=x.device)
    if extra_stream is not None:
        ws.record_stream(extra_stream)
        gw.record_stream(extra_stream)
    _hip.launch("conv3x3_wgrad", lambda: _hip.check(lib.savfi_conv3x3_wgrad_f32(
        x.data_ptr(), gz.data_ptr(), gw.data_ptr(), ws.data_ptr(), N, Ci, Co, H, W, int(pad),
        _hip.current_stream() if stream is None else stream), "savfi_conv3x3_wgrad_f32"), flops=18.0 * Ci * Co * gz.shape[2] * gz.shape[3] * N)
    return gw


def conv_bias_act(x, weight, bias, stride=1, padding=0, dilation=1, groups=1, slope=0.0, direct=False, cache=None, reflect=False,
                  in_slope=None, defer=False):
    """act(conv2d(x, weight) + bias) with act = LeakyReLU(slope) (0 -> ReLU, 1 -> identity); bias may be None.  `direct`: a 3x3
    layer wants the direct split-bf16 kernel whatever its size (no Winograd rounding); `cache`: a dict owned by the module whose
    own parameter `weight` is (its packed filters are kept there per weight version), None for fast weights.
    conv -> act -> conv chains whose intermediate has ONE consumer (model_utils.MetaSequential): `defer` = that consumer applies this
    layer's activation derivative, `in_slope` = x is the activated output of a producer that deferred its derivative to this layer
    (folded into this layer's data gradient: one element-wise pass over the map less per chain link)."""
    return _ConvBiasAct.apply(x, weight, bias, stride, padding, dilation, groups, float(slope), bool(direct), cache, bool(reflect),
                              None if in_slope is None else float(in_slope), bool(defer))


# --------------------------------------------------------------------------------------------
# Channel attention + residual of CAIN's RCAB        (reference model_utils.py:931-953, :957-990)
# --------------------------------------------------------------------------------------------
CA_FUSE_MLP = True       # A/B and tests: False = pool, MLP and apply as three launches per direction


class _ChannelAttentionResidual(torch.autograd.Function):
    """out = t * sigmoid(W2 relu(W1 mean_hw(t) + b1) + b2) + x, also returns the attention y [N,C,1,1] (not differentiable on
    its own).  w1 [T,Cr,C] / b1 [T,Cr] / w2 [T,C,Cr] / b2 [T,C]: sample n uses set n % T.  First-order only."""

    @staticmethod
    def forward(ctx, t, x, w1, b1, w2, b2):
        t, x = t.contiguous(), x.contiguous()
        w1, b1, w2, b2 = w1.contiguous(), b1.contiguous(), w2.contiguous(), b2.contiguous()
        _hip.require_cuda(t, x, w1, b1, w2, b2)
        N, C, H, W = t.shape
        T, Cr = w1.shape[0], w1.shape[1]
        assert x.shape == t.shape and N % T == 0 and tuple(w1.shape) == (T, Cr, C) and tuple(w2.shape) == (T, C, Cr), (t.shape, w1.shape, w2.shape)
        lib = _hip.lib()
        st = _hip.current_stream()
        s = torch.empty((N, C), dtype=t.dtype, device=t.device)
        y = torch.empty((N, C), dtype=t.dtype, device=t.device)
        a1 = torch.empty((N, Cr), dtype=t.dtype, device=t.device)
        out = torch.empty_like(t)
        hw = H * W
        _hip.launch("ca_pool", lambda: _hip.check(lib.savfi_ca_pool_f32(t.data_ptr(), None, s.data_ptr(), N * C, hw, 1.0 / hw, st),
                                                  "savfi_ca_pool_f32"), nbytes=4 * t.numel())
        if CA_FUSE_MLP and Cr <= 16 and C <= 256:       # the MLP inside the apply launch (every workgroup repeats it for its sample: the same y, bit for bit)
            _hip.launch("ca_apply", lambda: _hip.check(lib.savfi_ca_apply_mlp_f32(
                t.data_ptr(), s.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), x.data_ptr(), out.data_ptr(),
                y.data_ptr(), a1.data_ptr(), N, T, C, Cr, hw, st), "savfi_ca_apply_mlp_f32"), nbytes=12 * t.numel())
        else:
            _hip.launch("ca_mlp", lambda: _hip.check(lib.savfi_ca_mlp_fwd_f32(s.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                                                              y.data_ptr(), a1.data_ptr(), N, T, C, Cr, st), "savfi_ca_mlp_fwd_f32"))
            _hip.launch("ca_apply", lambda: _hip.check(lib.savfi_ca_apply_f32(t.data_ptr(), y.data_ptr(), x.data_ptr(), None, out.data_ptr(), N * C, hw, st),
                                                       "savfi_ca_apply_f32"), nbytes=12 * t.numel())
        ctx.save_for_backward(t, s, y, a1, w1, w2)
        ctx.mark_non_differentiable(y)
        ctx.set_materialize_grads(False)        # (else autograd fills a zero tensor for the attention's cotangent on every backward call)
        return out, y.view(N, C, 1, 1)

    @staticmethod
    @once_differentiable
    def backward(ctx, g, _gy):
        if g is None:
            return (None,) * 6
        t, s, y, a1, w1, w2 = ctx.saved_tensors
        g = g.contiguous()
        N, C, H, W = t.shape
        T, Cr = w1.shape[0], w1.shape[1]
        hw = H * W
        lib = _hip.lib()
        st = _hip.current_stream()
        r = torch.empty((N, C), dtype=t.dtype, device=t.device)
        ds = torch.empty((N, C), dtype=t.dtype, device=t.device)
        gw1, gb1 = torch.empty_like(w1), torch.empty((T, Cr), dtype=t.dtype, device=t.device)
        gw2, gb2 = torch.empty_like(w2), torch.empty((T, C), dtype=t.dtype, device=t.device)
        gt = torch.empty_like(t)
        _hip.launch("ca_pool", lambda: _hip.check(lib.savfi_ca_pool_f32(g.data_ptr(), t.data_ptr(), r.data_ptr(), N * C, hw, 1.0, st),
                                                  "savfi_ca_pool_f32"), nbytes=8 * t.numel())
        if CA_FUSE_MLP and Cr <= 16 and C <= 256:       # the MLP's backward inside the apply launch: ds per workgroup, the parameter gradients from T workgroups
            _hip.launch("ca_apply", lambda: _hip.check(lib.savfi_ca_apply_bwd_mlp_f32(             # at the front of its grid
                g.data_ptr(), r.data_ptr(), s.data_ptr(), y.data_ptr(), a1.data_ptr(), w1.data_ptr(), w2.data_ptr(), gt.data_ptr(), gw1.data_ptr(),
                gb1.data_ptr(), gw2.data_ptr(), gb2.data_ptr(), N, T, C, Cr, hw, st), "savfi_ca_apply_bwd_mlp_f32"), nbytes=8 * t.numel())
        else:
            _hip.launch("ca_mlp_bwd", lambda: _hip.check(lib.savfi_ca_mlp_bwd_f32(
                r.data_ptr(), s.data_ptr(), y.data_ptr(), a1.data_ptr(), w1.data_ptr(), w2.data_ptr(), ds.data_ptr(), gw1.data_ptr(), gb1.data_ptr(),
                gw2.data_ptr(), gb2.data_ptr(), N, T, C, Cr, 1.0 / hw, st), "savfi_ca_mlp_bwd_f32"))
            _hip.launch("ca_apply", lambda: _hip.check(lib.savfi_ca_apply_f32(g.data_ptr(), y.data_ptr(), None, ds.data_ptr(), gt.data_ptr(), N * C, hw, st),
                                                       "savfi_ca_apply_f32"), nbytes=8 * t.numel())
        need = ctx.needs_input_grad
        return (gt if need[0] else None, g if need[1] else None, gw1 if need[2] else None, gb1 if need[3] else None,
                gw2 if need[4] else None, gb2 if need[5] else None)


def channel_attention_residual(t, x, w1, b1, w2, b2):
    """RCAB tail: (t * CA(t) + x, CA(t)).  Weights of the two 1x1 convolutions as [Cr,C,1,1] / [Cr] / [C,Cr,1,1] / [C], or with a
    leading task axis T (tasks in lockstep: sample n uses set n % T)."""
    stacked = w1.dim() == 5
    T = w1.shape[0] if stacked else 1
    Cr, C = (w1.shape[1], w1.shape[2]) if stacked else (w1.shape[0], w1.shape[1])
    out, y = _ChannelAttentionResidual.apply(t, x, w1.reshape(T, Cr, C), b1.reshape(T, Cr), w2.reshape(T, C, Cr), b2.reshape(T, C))
    return out, y


# --------------------------------------------------------------------------------------------
# Bilinear x2 up-sampling      (sepconv/model.py:191, :213-234; voxel_flow.py:400-414)
# --------------------------------------------------------------------------------------------
class _Upsample2x(torch.autograd.Function):
    """geom = (H, W, sy0, sx0, Hs, Ws, oy0, ox0, Hw, Ww): x is the crop [sy0:sy0+Hs, sx0:sx0+Ws] of a virtual
    [H, W] map, the result the window [oy0:oy0+Hw, ox0:ox0+Ww] of its x2 up-sampling (full op: crop = window = all)."""

    @staticmethod
    def forward(ctx, x, align_corners, geom, in_slope=None):
        """in_slope: x is the activated output of a fused convolution that left its (leaky) ReLU derivative to this op (its single
        consumer; conv_bias_act `defer`): the adjoint multiplies by it in its store (first-order passes only)."""
        _hip.require_cuda(x)
        N, C = x.shape[:2]
        H, W, sy0, sx0, Hs, Ws, oy0, ox0, Hw, Ww = geom
        assert tuple(x.shape[2:]) == (Hs, Ws), (x.shape, geom)
        out = torch.empty((N, C, Hw, Ww), dtype=x.dtype, device=x.device)
        lib = _hip.lib()
        _hip.launch("upsample2x_fwd", lambda: _hip.check(lib.savfi_upsample2x_window_fwd_f32(
            x.data_ptr(), out.data_ptr(), N * C, *geom, int(align_corners), _hip.current_stream()),
            "savfi_upsample2x_window_fwd_f32"), nbytes=4 * N * C * (Hs * Ws + Hw * Ww))
        ctx.align, ctx.geom, ctx.in_slope = bool(align_corners), geom, in_slope
        if in_slope is not None:
            ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.in_slope is not None:
            # a raw kernel without a graph: under create_graph=True it would hand out zero second-order terms silently -- refuse instead
            # (the callers pass in_slope in first-order passes only: hip_ops.double_backward() is checked at forward time)
            if torch.is_grad_enabled() and g.requires_grad:
                raise RuntimeError("upsample2x with a deferred activation derivative (in_slope) is first-order only")
            x, = ctx.saved_tensors
            g = g.contiguous()
            N, C = g.shape[:2]
            H, W, sy0, sx0, Hs, Ws, oy0, ox0, Hw, Ww = ctx.geom
            gin = torch.empty((N, C, Hs, Ws), dtype=g.dtype, device=g.device)
            lib = _hip.lib()
            _hip.launch("upsample2x_bwd", lambda: _hip.check(lib.savfi_upsample2x_window_bwd_masked_f32(
                g.data_ptr(), x.data_ptr(), float(ctx.in_slope), gin.data_ptr(), N * C, *ctx.geom, int(ctx.align), _hip.current_stream()),
                "savfi_upsample2x_window_bwd_masked_f32"), nbytes=4 * N * C * (2 * Hs * Ws + Hw * Ww))
            return gin, None, None, None
        # linear op: its adjoint goes through the Function too, so double-backward keeps working
        return _Upsample2xAdjoint.apply(g, ctx.align, ctx.geom), None, None, None


class _Upsample2xAdjoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, align_corners, geom):
        g = g.contiguous()
        _hip.require_cuda(g)
        N, C = g.shape[:2]
        H, W, sy0, sx0, Hs, Ws, oy0, ox0, Hw, Ww = geom
        assert tuple(g.shape[2:]) == (Hw, Ww), (g.shape, geom)
        gin = torch.empty((N, C, Hs, Ws), dtype=g.dtype, device=g.device)
        lib = _hip.lib()
        _hip.launch("upsample2x_bwd", lambda: _hip.check(lib.savfi_upsample2x_window_bwd_f32(
            g.data_ptr(), gin.data_ptr(), N * C, *geom, int(align_corners), _hip.current_stream()),
            "savfi_upsample2x_window_bwd_f32"), nbytes=4 * N * C * (Hs * Ws + Hw * Ww))
        ctx.align, ctx.geom = bool(align_corners), geom
        return gin

    @staticmethod
    def backward(ctx, gg):
        return _Upsample2x.apply(gg.contiguous(), ctx.align, ctx.geom, None), None, None


def upsample_window_sources(o0, o1, size_in, align_corners):
    """First / last source index the outputs [o0, o1) of a x2 up-sampling of `size_in` samples read
    (float32 arithmetic of the kernel, which is ATen's area_pixel_compute_source_index)."""
    import numpy as np
    f = np.float32
    if align_corners:
        scale = f(size_in - 1) / f(2 * size_in - 1) if size_in > 0 else f(0)
        src = lambda d: scale * f(d)
    else:
        src = lambda d: max((f(d) + f(0.5)) * f(0.5) - f(0.5), f(0))
    lo = min(int(src(o0)), size_in - 1)
    hi = min(int(src(o1 - 1)), size_in - 1)
    return lo, min(hi + 1, size_in - 1)


def upsample_bilinear2x_window(x, full_hw, crop_origin, out_window, align_corners, in_slope=None):
    """x = crop of a virtual [N,C,*full_hw] map starting at crop_origin (y, x); returns rows/cols
    out_window = (oy0, ox0, Hw, Ww) of its bilinear x2 up-sampling.  in_slope: see _Upsample2x.forward."""
    H, W = full_hw
    geom = (int(H), int(W), int(crop_origin[0]), int(crop_origin[1]), int(x.shape[2]), int(x.shape[3]),
            int(out_window[0]), int(out_window[1]), int(out_window[2]), int(out_window[3]))
    return _Upsample2x.apply(x.contiguous(), bool(align_corners), geom, None if in_slope is None else float(in_slope))


def upsample_bilinear2x(x, align_corners, in_slope=None):
    """[N,C,H,W] -> [N,C,2H,2W], bilinear, ATen-identical source indices.  in_slope: see _Upsample2x.forward."""
    H, W = int(x.shape[2]), int(x.shape[3])
    return _Upsample2x.apply(x.contiguous(), bool(align_corners), (H, W, 0, 0, H, W, 0, 0, 2 * H, 2 * W),
                             None if in_slope is None else float(in_slope))


class Upsample2x(torch.nn.Module):
    """Parameter-free stand-in for torch.nn.Upsample(scale_factor=2, mode='bilinear', align_corners=...)."""

    def __init__(self, align_corners=True):
        super().__init__()
        self.align_corners = align_corners

    def forward(self, x, in_slope=None):
        if not x.is_cuda:    # CPU tensors (host-logic tests): the plain ATen op
            assert in_slope is None, "a deferred activation derivative is a GPU path"
            return torch.nn.functional.interpolate(x, scale_factor=2, mode='bilinear', align_corners=self.align_corners)
        return upsample_bilinear2x(x, self.align_corners, in_slope)

    def extra_repr(self):
        return 'scale_factor=2, mode=bilinear, align_corners=%s' % self.align_corners
