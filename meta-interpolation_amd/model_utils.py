"""Fast-weight ("Meta") layers shared by the SepConv / VoxelFlow / CAIN plugins.

Surface kept from the reference (model_utils.py:11-28, :202-228, :272-366, :821-1053): every Meta*
layer is an nn.Module whose ``forward(x, params=None)`` uses its own parameters when ``params`` is
None and externally supplied fast weights otherwise; parameter NAMES are part of the contract (they
key the inner-loop lr tables and checkpoints).

Routing is different by design.  The reference rebuilds nested dicts with ``extract_top_level_dict``
at every nesting level of every forward (O(#params x depth) Python work per pass, 494 keys x ~6
levels for CAIN).  Here the fast weights stay in ONE flat ``{name: tensor}`` dict for the whole inner
loop and layers receive a ``ParamView`` = (flat dict, name prefix); descending a level is a string
concatenation and a leaf lookup is a single dict access.  Plain (possibly nested) dicts are still
accepted anywhere a ParamView is, so the reference's call pattern ``module(x, params=subdict)`` works.
"""
import os
import threading

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip_ops


# --------------------------------------------------------------------------------------------
# fast-weight routing
# --------------------------------------------------------------------------------------------
class ParamView:
    """A window ``prefix*`` onto a flat fast-weight dict."""
    __slots__ = ("flat", "prefix")

    def __init__(self, flat, prefix=""):
        self.flat = flat
        self.prefix = prefix

    def sub(self, name):
        return ParamView(self.flat, self.prefix + str(name) + ".")

    def __getitem__(self, name):
        key = self.prefix + str(name)
        if key in self.flat:
            return self.flat[key]
        # an interior name: hand out the sub-window (mirrors params['conv'] returning a sub-dict)
        pre = key + "."
        if any(k.startswith(pre) for k in self.flat):
            return ParamView(self.flat, pre)
        raise KeyError(key)

    def __contains__(self, name):
        key = self.prefix + str(name)
        return key in self.flat or any(k.startswith(key + ".") for k in self.flat)

    def leaf(self, name):
        return self.flat[self.prefix + name]


def _flatten(d, prefix, out):
    for k, v in d.items():
        if isinstance(v, dict):
            _flatten(v, prefix + str(k) + ".", out)
        else:
            out[prefix + str(k)] = v
    return out


def as_view(params):
    """None | ParamView | flat dict | nested dict  ->  None | ParamView."""
    if params is None or isinstance(params, ParamView):
        return params
    if any(isinstance(v, dict) for v in params.values()):
        params = _flatten(params, "", {})
    return ParamView(params)


def extract_top_level_dict(current_dict):
    """Compatibility helper with the reference's name (model_utils.py:272-305): split a flat
    ``a.b.c`` dict on its first level.  The plugins in this package do not need it (see ParamView)."""
    out = {}
    for key, value in current_dict.items():
        name = key.replace("layer_dict.", "").replace("block_dict.", "").replace("module-", "")
        top, _, rest = name.partition(".")
        if rest == "":
            out[top] = value
        else:
            out.setdefault(top, {})[rest] = value
    return out


# --------------------------------------------------------------------------------------------
# small tensor helpers
# --------------------------------------------------------------------------------------------
def sub_mean(x):
    """Remove the per-channel spatial mean; returns (x - mean, mean)  (reference :11-15).  One savfi op (fixed summation order,
    safe inside captured hipGraphs -- ATen's large-frame reduction is not: csrc/submean.hip) for fp32 NCHW tensors on the GPU;
    anything else (CPU tensors of --cuda False / gloo runs, other dtypes) takes the reference's two ATen means."""
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 4:
        return hip_ops.sub_mean(x)
    mean = x.mean(2, keepdim=True).mean(3, keepdim=True)
    return x - mean, mean


def _pad_to_multiple(size, shift):
    return 0 if size == ((size >> shift) << shift) else (((size >> shift) + 1) << shift) - size


def InOutPaddings(x):
    """Reflection pad to a multiple of 128 and the matching crop (reference :17-28)."""
    pw, ph = _pad_to_multiple(x.size(3), 7), _pad_to_multiple(x.size(2), 7)
    left, top = pw // 2, ph // 2
    pad_in = nn.ReflectionPad2d([left, pw - left, top, ph - top])
    pad_out = nn.ReflectionPad2d([-left, left - pw, -top, top - ph])
    return pad_in, pad_out


def pixel_shuffle(input, scale_factor):
    """Space-to-depth (scale<1) / depth-to-space (scale>=1) with the reference's channel order
    (reference :202-217), on the savfi HIP permutation kernels."""
    return hip_ops.pixel_shuffle(input, scale_factor)


class PixelShuffle(nn.Module):
    def __init__(self, scale_factor):
        super().__init__()
        self.scale_factor = scale_factor

    def forward(self, x):
        return pixel_shuffle(x, self.scale_factor)

    def extra_repr(self):
        return 'scale_factor={}'.format(self.scale_factor)


# --------------------------------------------------------------------------------------------
# Meta layers
# --------------------------------------------------------------------------------------------
# conv -> bias -> (Leaky)ReLU runs with fused epilogue kernels (hip_ops.conv_bias_act, --fuse_conv_act 1, the default):
# bias add + clamp become one in-place pass over the conv output, threshold_backward + the per-channel bias-gradient
# sum one pass over the cotangent (76.9 -> 82.4 inner steps/s on SepConv 256x448, eager).  The fused backward is
# first-order only (off under --second_order).  Round-1 note: this was slower at first (58.7 vs 62.3) because a
# custom Function cannot tell that autograd.grad() does not need its weight gradient - see OWN_PARAMS_CONST.
# (per thread, like OWN_PARAMS_CONST below: set_fuse_conv_act / fuse_conv_act)
# True while a first-order support pass runs: layers that use their OWN parameters (not the fast-weight dict) treat
# them as constants.  The inner gradient is taken w.r.t. the fast weights only and its graph is dropped, so nothing
# changes - but custom autograd Functions cannot see which of their inputs a particular autograd.grad() call
# needs (ctx.needs_input_grad only says requires_grad) and would compute unused weight gradients.
# Per THREAD: tasks may be adapted concurrently, one Python thread and HIP stream each (meta_learning_system.py).
_TLS = threading.local()


def fuse_conv_act():
    return getattr(_TLS, 'fuse_conv_act', False)


def set_fuse_conv_act(value):
    _TLS.fuse_conv_act = bool(value)


_FUSE_CONV_CHAIN = True      # A/B and tests: model_utils._FUSE_CONV_CHAIN = False switches every deferral off


def fuse_conv_chain():
    """conv -> act -> conv inside a MetaSequential: the consumer folds the producer's activation derivative into its data gradient
    (hip_ops.conv_bias_act `defer` / `in_slope`).  SAVFI_NO_CONV_CHAIN=1 keeps every layer's own element-wise pass (A/B runs)."""
    return _FUSE_CONV_CHAIN


def own_params_const():
    return getattr(_TLS, 'own_params_const', False)


def set_own_params_const(value):
    _TLS.own_params_const = bool(value)


def _act_slope(module):
    """Negative slope if `module` is an activation the fused epilogue implements, else None."""
    if isinstance(module, nn.ReLU):
        return 0.0
    if isinstance(module, nn.LeakyReLU):
        return float(module.negative_slope)
    return None


class MetaConv2dLayer(nn.Module):
    """conv2d with internal (Xavier-uniform weight, zero bias) or external weights (reference :308-366)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, use_bias=True, groups=1,
                 dilation_rate=1, direct=False):
        super().__init__()
        # direct = True: a 3x3 layer of a network that amplifies Winograd rounding (VoxelFlow) asks for the direct split-bf16
        # convolution whatever its size; 5x5 / 7x7 layers take it anyway (hip_ops.convk_eligible)
        self.direct = bool(direct)
        self._filters = {}          # packed / transformed filters of self.weight (hip_ops._filters)
        self.stride, self.padding = int(stride), int(padding)
        self.dilation_rate, self.groups, self.use_bias = int(dilation_rate), int(groups), use_bias
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        nn.init.xavier_uniform_(self.weight)
        self.bias = nn.Parameter(torch.zeros(out_channels)) if use_bias else None

    def forward(self, x, params=None, act_slope=None, padding=None, reflect=False, in_slope=None, chain=None):
        """`act_slope` (set by MetaSequential when an activation follows) applies LeakyReLU(act_slope);
        `padding` overrides the layer's own zero padding (windowed evaluation, sepconv/model.py); `reflect`: that padding mirrors
        the image (MetaConvNorm; the caller has checked hip_ops.convk_reflect_eligible).
        conv -> act -> conv chains (MetaSequential): `chain` = {'want_defer': True} asks this layer to leave its activation derivative
        to its single consumer -- honoured on the fused paths only, which then set chain['deferred'] --; `in_slope` = x is the
        activated output of a producer that did: its derivative is folded into this layer's data gradient."""
        padding = self.padding if padding is None else padding
        want_defer = bool(chain and chain.get('want_defer')) and act_slope is not None and not reflect

        def fused(fn, *a):
            if want_defer:
                chain['deferred'] = True
            return fn(*a, in_slope=in_slope, defer=want_defer)
        if params is not None:
            pv = as_view(params)
            weight = pv.leaf("weight")
            bias = pv.leaf("bias") if self.use_bias else None
        elif own_params_const():
            weight, bias = self.weight.detach(), (self.bias.detach() if self.bias is not None else None)
        else:
            weight, bias = self.weight, self.bias
        direct = getattr(self, 'direct', False)
        if weight.dim() == 5:
            # fast weights stacked over the tasks of a meta-batch adapted in lockstep: [T, Co, Ci, kh, kw], x [n*T, ...]
            if x.is_cuda and fuse_conv_act() and self.groups == 1 and (
                    act_slope is not None or bias is None or hip_ops.conv3x3_tasks_eligible(x, weight, self.stride, padding, self.dilation_rate)
                    or hip_ops.convk_eligible(x, weight, self.stride, padding, self.dilation_rate, 1, direct)):
                return fused(hip_ops.conv_bias_act_tasks, x, weight, bias, self.stride, padding, self.dilation_rate,
                             1.0 if act_slope is None else act_slope, direct)
            assert self.groups == 1, "lockstep tasks on a grouped convolution"
            if in_slope is not None:
                x = hip_ops.mask_grad(x, in_slope)
            out = hip_ops.conv2d_tasks(x, weight, bias, self.stride, padding, self.dilation_rate)
            if act_slope is not None:
                out = F.relu(out) if act_slope == 0.0 else F.leaky_relu(out, act_slope)
            return out
        if x.is_cuda and fuse_conv_act():
            own = self._filters if params is None else None     # own parameter: packed filters cached per weight version, here
            if hip_ops.convk_eligible(x, weight, self.stride, padding, self.dilation_rate, self.groups, direct):
                return fused(hip_ops.conv_bias_act, x, weight, bias, self.stride, padding, self.dilation_rate, self.groups,
                             1.0 if act_slope is None else act_slope, direct, own, reflect)
            assert not reflect, "mirrored borders are a direct-kernel path (hip_ops.convk_reflect_eligible)"
            if bias is not None and act_slope is not None:
                return fused(hip_ops.conv_bias_act, x, weight, bias, self.stride, padding, self.dilation_rate, self.groups,
                             act_slope, direct, own)
            if bias is not None and hip_ops.conv3x3_eligible(x, weight, self.stride, padding, self.dilation_rate, self.groups):
                # no activation follows: still worth the savfi kernel (bias in its epilogue) for large maps
                return fused(hip_ops.conv_bias_act, x, weight, bias, self.stride, padding, self.dilation_rate, self.groups, 1.0, direct, own)
        assert not reflect, "mirrored borders are a direct-kernel path (hip_ops.convk_reflect_eligible)"
        if in_slope is not None:            # not a fused path: the deferred derivative as an identity node on the input
            x = hip_ops.mask_grad(x, in_slope)
        out = F.conv2d(x, weight, bias, self.stride, padding, self.dilation_rate, self.groups)
        if act_slope is not None:
            out = F.relu(out) if act_slope == 0.0 else F.leaky_relu(out, act_slope)
        return out

    def restore_backup_stats(self):
        pass


def conv_pair(conv_a, conv_b, x, params_a, params_b, slope_a, slope_b=None):
    """conv_a -> (Leaky)ReLU(slope_a) -> conv_b [-> (Leaky)ReLU(slope_b)] for two MetaConv2dLayers where conv_a's activated output has
    conv_b as its ONLY consumer (the blocks of the RRIN / Super SloMo UNets, written out by hand there): MetaSequential's conv -> act ->
    conv protocol -- conv_b's data gradient applies conv_a's activation derivative in its epilogue, conv_a's bias gradient rides on its
    weight gradient: no element-wise pass over conv_a's cotangent."""
    chain = ({"want_defer": True} if (x.is_cuda and fuse_conv_chain() and fuse_conv_act() and not hip_ops.double_backward()
                                      and torch.is_grad_enabled()) else None)
    y = conv_a(x, params=params_a, act_slope=slope_a, **({} if chain is None else {"chain": chain}))
    kw = {"in_slope": slope_a} if (chain is not None and chain.get("deferred")) else {}
    if slope_b is not None:
        kw["act_slope"] = slope_b
    return conv_b(y, params=params_b, **kw)


class MetaConvNorm(nn.Module):
    """reflection pad + conv (norm is never enabled by the three plugins; reference :821-848)."""

    def __init__(self, in_feat, out_feat, kernel_size, stride=1, norm=False):
        super().__init__()
        assert not norm, "normalisation inside MetaConvNorm is unused on this path"
        self.reflection_pad = nn.ReflectionPad2d(kernel_size // 2)
        self.conv = MetaConv2dLayer(in_feat, out_feat, kernel_size=kernel_size, stride=stride, padding=0,
                                    use_bias=True)
        self.norm = norm

    def forward(self, x, params=None, act_slope=None, in_slope=None, chain=None, want_skip=False):
        """`chain` / `in_slope`: MetaConv2dLayer's conv -> act -> conv protocol, through the mirrored border: a padded map's ReLU mask is
        the mirrored mask, so the consumer's data gradient masked by its PADDED input and then folded is d/dz of the producer.
        `want_skip`: returns (result, x') with x' = x for a connection round this layer (an RCAB's skip): where a padded copy is made,
        the cotangent that comes back along x' is added inside the pad's adjoint (hip_ops.reflect_pad_with_skip)."""
        pv = as_view(params)
        sub = None if pv is None else pv.sub("conv")
        pad = self.reflection_pad.padding[0]
        if x.is_cuda and fuse_conv_act() and not _NO_REFLECT_FUSED:
            # the direct kernel mirrors the border while it stages its tile: no padded copy of x, forward or backward
            weight = self.conv.weight if sub is None else as_view(sub).leaf("weight")
            if hip_ops.convk_reflect_eligible(x, weight, pad):
                out = self.conv(x, params=sub, act_slope=act_slope, padding=pad, reflect=True, in_slope=in_slope)
                return (out, x) if want_skip else out
        xp, xs = hip_ops.reflect_pad_with_skip(x, pad) if want_skip else (hip_ops.reflect_pad(x, pad), x)
        out = self.conv(xp, params=sub, act_slope=act_slope, in_slope=in_slope, chain=chain)
        return (out, xs) if want_skip else out


_META_TYPES = ()
_NO_CA_FUSED = False           # A/B (module attributes): True = CAIN's channel attention / mirrored borders from their unfused ops
_NO_REFLECT_FUSED = False


class MetaSequential(nn.Sequential):
    """nn.Sequential whose meta children receive ``params[str(index)]`` (reference :851-891)."""

    def is_meta_layer(self, module):
        return isinstance(module, _META_TYPES)

    def forward(self, input, params=None, in_slope=None, defer_last=False):
        """in_slope: `input` is the activated output of a convolution that left its activation derivative to this Sequential's first
        module (a MetaConv2dLayer or an Upsample2x).  defer_last: the caller promises that the result has exactly one consumer that can
        take over the LAST conv + activation pair's derivative (an Upsample2x; a Sequential called with in_slope): returns (result,
        slope or None) instead of the result."""
        pv = as_view(params)
        mods = list(self)
        ind = 0                         # in_slope: the previous conv left its activation derivative to the next module (see below)
        while ind < len(mods):
            module = mods[ind]
            kw, step = {}, 1
            if isinstance(module, (MetaConv2dLayer, MetaConvNorm)) and ind + 1 < len(mods):
                slope = _act_slope(mods[ind + 1])
                if slope is not None:          # conv + activation pair: one fused call, skip the activation module
                    kw["act_slope"] = slope
                    step = 2
            chain = None
            if isinstance(module, MetaConv2dLayer):
                if in_slope is not None:
                    kw["in_slope"] = in_slope
                # conv -> act -> conv: the intermediate map has exactly one consumer (the next conv of this Sequential), which can fold
                # this layer's activation derivative into its data gradient -- first-order GPU passes only
                # (round 5: ... or an Upsample2x, whose adjoint multiplies by it in its store; or, with defer_last, the caller's consumer)
                takes_over = (isinstance(mods[ind + 2], (MetaConv2dLayer, hip_ops.Upsample2x)) if ind + 2 < len(mods) else bool(defer_last))
                if "act_slope" in kw and takes_over and input.is_cuda \
                        and fuse_conv_chain() and fuse_conv_act() and not hip_ops.double_backward() and torch.is_grad_enabled():
                    chain = {"want_defer": True}
                    kw["chain"] = chain
            elif isinstance(module, hip_ops.Upsample2x):
                if in_slope is not None:
                    kw["in_slope"] = in_slope
            elif in_slope is not None:       # (a first module that cannot take it over: the derivative as an identity node on the input)
                input = hip_ops.mask_grad(input, in_slope)
            if pv is not None and isinstance(module, _META_TYPES):
                input = module(input, params=pv.sub(ind), **kw)
            else:
                input = module(input, **kw)
            in_slope = kw["act_slope"] if (chain is not None and chain.get("deferred")) else None
            ind += step
        if defer_last:
            return input, in_slope
        assert in_slope is None
        return input

    def restore_backup_stats(self):
        pass


class MetaCALayer(nn.Module):
    """Channel attention: GAP -> 1x1 (C -> C/r) -> ReLU -> 1x1 -> sigmoid -> scale (reference :931-953)."""

    def __init__(self, channel, reduction=16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.conv_du = MetaSequential(
            MetaConv2dLayer(channel, channel // reduction, kernel_size=1, stride=1, padding=0),
            nn.ReLU(inplace=False),
            MetaConv2dLayer(channel // reduction, channel, kernel_size=1, stride=1, padding=0),
            nn.Sigmoid())

    def forward(self, x, params=None):
        pv = as_view(params)
        y = self.conv_du(self.avg_pool(x), None if pv is None else pv.sub("conv_du"))
        return x * y, y


class MetaRCAB(nn.Module):
    """Residual channel-attention block (reference :957-990); no downscale variant on this path."""

    def __init__(self, in_feat, out_feat, kernel_size, reduction, bias=True, norm=False, act=nn.ReLU(True),
                 downscale=False, return_ca=False):
        super().__init__()
        assert not downscale, "downscaling RCABs are not used by MetaCAIN"
        self.body = MetaSequential(
            MetaConvNorm(in_feat, out_feat, kernel_size, stride=1, norm=norm),
            act,
            MetaConvNorm(out_feat, out_feat, kernel_size, stride=1, norm=norm),
            MetaCALayer(out_feat, reduction))
        self.downscale = downscale
        self.return_ca = return_ca

    def forward(self, x, params=None):
        pv = as_view(params)
        if x.is_cuda and fuse_conv_act() and _act_slope(self.body[1]) is not None and not _NO_CA_FUSED:
            # first-order pass on the GPU: conv + LeakyReLU, conv, then pool -> MLP -> scale -> skip as the fused savfi op
            # (hip_ops.channel_attention_residual: three launches instead of eight, each map read once per launch)
            sub = (lambda i: None) if pv is None else (lambda i: pv.sub("body").sub(i))
            # conv -> ReLU -> (mirror) -> conv: the ReLU's derivative goes into the second convolution's data gradient (its masked
            # epilogue), the first convolution's bias gradient then rides on its weight-gradient kernel: no pass over the cotangent
            slope = _act_slope(self.body[1])
            chain = ({"want_defer": True} if (fuse_conv_chain() and not hip_ops.double_backward() and torch.is_grad_enabled()
                                              and isinstance(self.body[0], MetaConvNorm) and isinstance(self.body[2], MetaConvNorm))
                     else None)
            if isinstance(self.body[0], MetaConvNorm):      # (x's two cotangents -- this convolution's and the skip's -- meet in the pad's adjoint)
                t, x = self.body[0](x, params=sub(0), act_slope=slope, want_skip=True, **({} if chain is None else {"chain": chain}))
            else:
                t = self.body[0](x, params=sub(0), act_slope=slope)
            t = self.body[2](t, params=sub(2), **({"in_slope": slope} if (chain is not None and chain.get("deferred")) else {}))
            du = self.body[3].conv_du
            if pv is not None:
                leaf = pv.sub("body").sub(3).sub("conv_du")
                w1, b1, w2, b2 = leaf.sub(0).leaf("weight"), leaf.sub(0).leaf("bias"), leaf.sub(2).leaf("weight"), leaf.sub(2).leaf("bias")
            else:
                w1, b1, w2, b2 = du[0].weight, du[0].bias, du[2].weight, du[2].bias
                if own_params_const():
                    w1, b1, w2, b2 = w1.detach(), b1.detach(), w2.detach(), b2.detach()
            out, ca = hip_ops.channel_attention_residual(t, x, w1, b1, w2, b2)
            return (out, ca) if self.return_ca else out
        out, ca = self.body(x, None if pv is None else pv.sub("body"))
        out = out + x
        return (out, ca) if self.return_ca else out


class MetaResidualGroup(nn.Module):
    """n_resblocks blocks + one conv, with a group-level skip (reference :994-1011)."""

    def __init__(self, Block, n_resblocks, n_feat, kernel_size, reduction, act, norm=False):
        super().__init__()
        blocks = [Block(n_feat, n_feat, kernel_size, reduction, bias=True, norm=norm, act=act)
                  for _ in range(n_resblocks)]
        blocks.append(MetaConvNorm(n_feat, n_feat, kernel_size, stride=1, norm=norm))
        self.body = MetaSequential(*blocks)

    def forward(self, x, params=None):
        pv = as_view(params)
        return self.body(x, None if pv is None else pv.sub("body")) + x


class MetaInterpolation(nn.Module):
    """CAIN trunk: head conv (2F -> F), residual groups, long skip, tail conv (reference :1014-1053)."""

    def __init__(self, n_resgroups, n_resblocks, n_feats, reduction=16, act=nn.LeakyReLU(0.2, False),
                 norm=False):
        super().__init__()
        self.headConv = MetaConv2dLayer(n_feats * 2, n_feats, kernel_size=3, stride=1, padding=1)
        self.body = MetaSequential(*[
            MetaResidualGroup(MetaRCAB, n_resblocks=n_resblocks, n_feat=n_feats, kernel_size=3,
                              reduction=reduction, act=act, norm=norm) for _ in range(n_resgroups)])
        self.tailConv = MetaConv2dLayer(n_feats, n_feats, kernel_size=3, stride=1, padding=1)

    def forward(self, x0, x1, params=None):
        pv = as_view(params)
        sub = (lambda n: None) if pv is None else pv.sub
        x = self.headConv(torch.cat([x0, x1], dim=1), sub("headConv"))
        res = self.body(x, sub("body")) + x
        return self.tailConv(res, sub("tailConv"))


_META_TYPES = (MetaConv2dLayer, MetaConvNorm, MetaSequential, MetaCALayer, MetaRCAB, MetaResidualGroup)


def zero_grad_params(module, params=None):
    """``net.zero_grad(params)`` of the plugins (e.g. sepconv/model.py:352-367) without the
    reference's per-parameter host sync (``torch.sum(param.grad) > 0`` + print)."""
    it = module.parameters() if params is None else params.values()
    for p in it:
        if p.is_leaf and p.requires_grad and p.grad is not None:
            p.grad = None
