"""SepConv plugin (``--model sepconv``): the adaptive separable-convolution U-Net with fast weights.

Surface and parameter names follow the reference's ``MetaNetwork`` (sepconv/model.py:168-375):
``forward(frame0, frame1, params=None, **kwargs) -> [N,3,H,W]``, ``zero_grad(params)``,
``restore_backup_stats()``; parameters ``moduleConv{1..5}.{0,2,4}``, ``moduleDeconv{5..2}.{0,2,4}``,
``moduleUpsample{5..2}.1``, ``module{Vertical,Horizontal}{1,2}.{0,2,4,7}`` (.weight/.bias).

As in the reference (:276-306 vs :292-307, :346-347) only the encoder/decoder ``Basic`` blocks read
the fast-weight dict; the four ``moduleUpsampleN`` and the four 51-tap ``Subnet``s always use the
module's own parameters.  The two local separable convolutions run on the savfi HIP kernel
(FunctionSepconv); frames are replication-padded by 25 px and up to a multiple of 128.

Windowed sub-networks (``windowed=True``, GPU tensors).  The reference evaluates the four Subnets and the
two separable convolutions on the whole padded canvas (384x512 for a 256x448 frame) and then keeps only the
frame area (:346-349) - 42 % of those pixels are discarded, and the Subnets are 52 % of the network's MACs.
Every operator after ``tensorCombine`` is local (3x3 convs, ReLU, bilinear x2, the per-pixel 51-tap op), so
the kept pixels depend only on a window of ``tensorCombine``: here the Subnets run on that window (+3 px
halo for their three half-resolution convs, +1 px for the full-resolution one), the up-sampling produces
just the window it feeds (savfi_upsample2x_window_*), and FunctionSepconv produces the H x W frame
directly from the frame padded by 25 px.  Same values as the full-canvas evaluation (which stays available:
``windowed=False`` / ``--sepconv_window 0``; tests compare the two), forward and backward.
"""
import os
import threading

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip_ops
from ..hip_ops import Upsample2x, upsample_bilinear2x_window, upsample_window_sources
from ..model_utils import MetaConv2dLayer, MetaSequential, as_view, fuse_conv_act, fuse_conv_chain, own_params_const, zero_grad_params
from .sepconv_op.sepconv import FunctionSepconv, FunctionSepconvPair, frames8_supported

TAPS_UNIT16 = True        # A/B (module attributes, no environment variable): False = planar taps / tap gradients
GRADS_UNIT16 = True
FRAME_CACHE = True        # False: pads / concatenation of the frame pair once per forward
CROP_EXACT = False        # True: the Subnets' window without the alignment slack

FILTER_TAPS = 51
HALF = FILTER_TAPS // 2  # 25


def _conv(cin, cout):
    return MetaConv2dLayer(in_channels=cin, out_channels=cout, kernel_size=3, stride=1, padding=1)


def _basic(cin, cout):
    return MetaSequential(_conv(cin, cout), nn.ReLU(inplace=False),
                          _conv(cout, cout), nn.ReLU(inplace=False),
                          _conv(cout, cout), nn.ReLU(inplace=False))


def _upsample(ch):
    return MetaSequential(Upsample2x(align_corners=True), _conv(ch, ch), nn.ReLU(inplace=False))


def _subnet():
    return MetaSequential(_conv(64, 64), nn.ReLU(inplace=False),
                          _conv(64, 64), nn.ReLU(inplace=False),
                          _conv(64, FILTER_TAPS), nn.ReLU(inplace=False),
                          Upsample2x(align_corners=True),
                          _conv(FILTER_TAPS, FILTER_TAPS))


_FRAME_TLS = threading.local()       # MetaNetwork._prepared_frames: per thread, per network

_ENCODER = [("moduleConv1", 6, 32), ("moduleConv2", 32, 64), ("moduleConv3", 64, 128),
            ("moduleConv4", 128, 256), ("moduleConv5", 256, 512)]
_DECODER = [("moduleDeconv5", 512, 512), ("moduleDeconv4", 512, 256), ("moduleDeconv3", 256, 128),
            ("moduleDeconv2", 128, 64)]


class MetaNetwork(nn.Module):
    lockstep_tasks = True     # verified against the sequential loop and the reference fixtures (tests/test_system_gpu.py)
    def __init__(self, resume=False, strModel='lf', windowed=True):
        super().__init__()
        self.windowed = bool(windowed)
        self.batch_subnets = True     # windowed tail: the four Subnets as one launch per layer (A/B: set False on the instance)
        self._windows = {}
        for i, (name, cin, cout) in enumerate(_ENCODER, start=1):
            setattr(self, name, _basic(cin, cout))
            setattr(self, "modulePool%d" % i, hip_ops.AvgPool2x2())
        for name, cin, cout in _DECODER:
            setattr(self, name, _basic(cin, cout))
            setattr(self, name.replace("Deconv", "Upsample"), _upsample(cout))
        self.moduleVertical1 = _subnet()
        self.moduleVertical2 = _subnet()
        self.moduleHorizontal1 = _subnet()
        self.moduleHorizontal2 = _subnet()
        self.padding = [HALF] * 4
        self.modulePad = nn.ReplicationPad2d(self.padding)
        if resume:
            path = 'pretrained_models/sepconv_base_' + strModel + '.pth'
            print('Loading model: ' + path)
            self.load_state_dict(torch.load(path, map_location='cpu', weights_only=False))

    @staticmethod
    def padded_size(height, width):
        """(H+50, W+50) rounded up to multiples of 128 (reference :254-260)."""
        up = lambda n: n if n == ((n >> 7) << 7) else (((n >> 7) + 1) << 7)
        return up(height + 2 * HALF), up(width + 2 * HALF)

    def forward(self, tensorFirst, tensorSecond, params=None, **kwargs):
        height, width = tensorFirst.size(2), tensorFirst.size(3)
        ph, pw = self.padded_size(height, width)
        prep = self._prepared_frames(tensorFirst, tensorSecond, ph, pw)
        first, second = prep['first'], prep['second']

        pv = as_view(params)
        fast = (lambda n: None) if pv is None else pv.sub

        x = prep['x']
        skips = []
        for i, (name, _, _) in enumerate(_ENCODER, start=1):
            # the block's activated output feeds the pooling and a skip connection: one op with one element-wise pass in backward (the
            # pooling's adjoint + the sum of the two cotangents + the block's last ReLU derivative, which the block leaves to it)
            x, slope = getattr(self, name)(x, fast(name), defer_last=True)
            if x.is_cuda:
                x, skip = hip_ops.avg_pool2x2_and_skip(x, slope)
            else:
                assert slope is None
                x, skip = getattr(self, "modulePool%d" % i)(x), x
            skips.append(skip)
        for name, _, _ in _DECODER:
            # (the block's last ReLU has one consumer, the bilinear x2 of its Upsample: its derivative is left to that op's adjoint)
            x, slope = getattr(self, name)(x, fast(name), defer_last=True)
            x = getattr(self, name.replace("Deconv", "Upsample"))(x, in_slope=slope)   # own parameters, as the reference
            x = x + skips.pop()
        combine = x  # [N,64,ph/2,pw/2]

        if self.windowed and combine.is_cuda:
            return self._windowed_tail(tensorFirst, tensorSecond, combine, height, width, ph, pw, prep)
        dot1 = FunctionSepconv.apply(self.modulePad(first).contiguous(),
                                     self.moduleVertical1(combine), self.moduleHorizontal1(combine))
        dot2 = FunctionSepconv.apply(self.modulePad(second).contiguous(),
                                     self.moduleVertical2(combine), self.moduleHorizontal2(combine))
        out = dot1 + dot2
        return out[:, :, HALF:HALF + height, HALF:HALF + width]

    # ---- what a forward makes of the two frames alone: the network's input canvas and the frames with the 51-tap op's rim ----
    # The inner loop runs every step on the SAME support frames (reference meta_learning_system.py:387-396: the task's support triplets),
    # so the four replication pads and the concatenation of a pass are made once per frame pair, not once per step: the last pair(s) are
    # kept per thread, keyed by the tensors' identity and version (the cache holds the tensors, so an address cannot come back as another
    # tensor).  Frames that carry gradients, CPU tensors and passes inside a hipGraph capture take the plain ops.
    def _prepared_frames(self, f0, f1, ph, pw):
        height, width = f0.size(2), f0.size(3)
        pad_in = (HALF, pw - HALF - width, HALF, ph - HALF - height)

        def make():
            first = F.pad(f0, pad_in, mode='replicate')
            second = F.pad(f1, pad_in, mode='replicate')
            out = dict(first=first, second=second, x=torch.cat([first, second], 1))
            if self.windowed and f0.is_cuda:
                rim = (HALF,) * 4
                out['rim0'], out['rim1'] = F.pad(f0, rim, mode='replicate'), F.pad(f1, rim, mode='replicate')
            return out
        if (not f0.is_cuda or f0.requires_grad or f1.requires_grad or torch.cuda.is_current_stream_capturing()
                or not FRAME_CACHE):
            return make()
        cache = _FRAME_TLS.__dict__.setdefault('entries', {}).setdefault(id(self), [])
        for e in cache:
            if e[0] is f0 and e[1] is f1 and e[2] == (f0._version, f1._version, ph, pw, torch.cuda.current_stream().cuda_stream):
                return e[3]
        with torch.no_grad():
            made = make()
        cache.append((f0, f1, (f0._version, f1._version, ph, pw, torch.cuda.current_stream().cuda_stream), made))
        del cache[:-2]                       # support frames + target frames of the current task group
        return made

    # ---- windowed evaluation of everything after tensorCombine (see the module docstring) -------------
    def _window(self, height, width, ph, pw):
        key = (height, width)
        if key not in self._windows:
            hh, hw = ph // 2, pw // 2
            # full-resolution window the last 3x3 conv reads: frame area +-1 (never touches the canvas border,
            # ph >= height + 2*HALF)
            up = (HALF - 1, HALF - 1, height + 2, width + 2)
            sy = upsample_window_sources(up[0], up[0] + up[2], hh, True)      # half-res rows / cols it reads
            sx = upsample_window_sources(up[1], up[1] + up[3], hw, True)
            # +3 halo for the three half-resolution convs; where the crop is clipped at the canvas border the
            # convs' own zero padding is the true one, elsewhere the (wrong) outer rings are never read
            cy0, cy1 = max(0, sy[0] - 3), min(hh, sy[1] + 1 + 3)
            cx0, cx1 = max(0, sx[0] - 3), min(hw, sx[1] + 1 + 3)
            # a width that is a multiple of 4 where the canvas allows (233 -> 236 at 256 x 448): the convolution epilogues and the
            # element-wise kernels store 16 bytes per lane instead of four dwords; the extra columns lie beyond the halo and are never read
            if not CROP_EXACT:
                extra = (-(cx1 - cx0)) % 4
                grow_r = min(extra, hw - cx1)
                cx1 += grow_r
                cx0 -= min(extra - grow_r, cx0)
            self._windows[key] = dict(half=(hh, hw), crop=(cy0, cy1, cx0, cx1), up=up)
        return self._windows[key]

    @staticmethod
    def _chain(x):
        """conv -> ReLU -> conv outside a MetaSequential (the windowed Subnets call their layers one by one): may the producer leave its
        activation derivative to its single consumer?  The rule of MetaSequential.forward (model_utils.py)."""
        return bool(x.is_cuda and fuse_conv_chain() and fuse_conv_act() and not hip_ops.double_backward() and torch.is_grad_enabled())

    def _subnet_window(self, seq, crop, win):
        ch = self._chain(crop)
        c0, c2 = ({'want_defer': True} if ch else None), ({'want_defer': True} if ch else None)
        slope = lambda c: 0.0 if (c is not None and c.get('deferred')) else None
        x = seq[0](crop, act_slope=0.0, chain=c0)
        x = seq[2](x, act_slope=0.0, in_slope=slope(c0), chain=c2)
        c4 = {'want_defer': True} if ch else None
        x = seq[4](x, act_slope=0.0, in_slope=slope(c2), chain=c4)
        x = upsample_bilinear2x_window(x, win['half'], (win['crop'][0], win['crop'][2]), win['up'], True, slope(c4))
        return seq[7](x, padding=0)       # [N,51,height,width]: exactly the frame area

    # ---- the four Subnets as ONE task-batched launch per layer (round 4) -------------------------------
    # The Subnets share their input and their layer shapes.  Layer 1 of all four is one 64 -> 256 convolution; its output
    # [N, 4 * 64, h, w] IS [4 N, 64, h, w] with sample 4 n + s belonging to Subnet s, the layout of the lockstep machinery (sample i uses
    # filter set i % T, hip_ops.conv_bias_act_tasks), so layers 2..4 run with T = 4 filter sets on it; the taps stay in that interleaved
    # buffer and FunctionSepconvPair reads them (and writes their gradients) in place.  5 convolution launches per pass instead of 16, whole
    # rounds of the chip on the 137 x 233 maps (4 608 workgroups = 9.0 rounds where one Subnet's 1 152 were 2.25), the data gradient of
    # layer 1 sums over the Subnets inside the convolution (no accumulation passes over `combine`'s gradient).  Same arithmetic per output
    # as the Subnet-by-Subnet evaluation (`batch_subnets=False`, SAVFI_SEPCONV_SUBNETS_ONE_BY_ONE=1: tests compare the two).
    _SUBNETS = ("moduleVertical1", "moduleHorizontal1", "moduleVertical2", "moduleHorizontal2")

    def _stacked_subnet_params(self):
        subs = [getattr(self, n) for n in self._SUBNETS]
        const = own_params_const()
        if const and torch.cuda.is_current_stream_capturing():
            # inside a hipGraph capture the stacking kernels belong to the graph (replayed on the live parameters); a cached tensor made
            # outside would be baked in by address and go stale with the next outer step
            with torch.no_grad():
                cat = lambda i, a: torch.cat([getattr(s[i], a).detach() for s in subs], 0)
                stack = lambda i, a: torch.stack([getattr(s[i], a).detach() for s in subs], 0)
                return dict(w0=cat(0, 'weight'), b0=cat(0, 'bias'), **{k + str(i): stack(i, a) for i in (2, 4, 7) for k, a in (('w', 'weight'), ('b', 'bias'))})
        if not const:                                  # the outer pass differentiates them: built inside the graph, every time
            cat = lambda i, a: torch.cat([getattr(s[i], a) for s in subs], 0)
            stack = lambda i, a: torch.stack([getattr(s[i], a) for s in subs], 0)
            return dict(w0=cat(0, 'weight'), b0=cat(0, 'bias'), **{k + str(i): stack(i, a) for i in (2, 4, 7) for k, a in (('w', 'weight'), ('b', 'bias'))})
        ps = [p for s in subs for i in (0, 2, 4, 7) for p in (s[i].weight, s[i].bias)]
        key = tuple((p.data_ptr(), p._version) for p in ps)
        stream = torch.cuda.current_stream().cuda_stream            # one copy per stream: task streams never share a tensor made on another
        cache = self.__dict__.setdefault('_subnet_stacks', {})
        hit = cache.get(stream)
        if hit is None or hit[0] != key:
            if hit is not None:
                for t in hit[1].values():
                    hip_ops.unregister_const_weight(t)
            with torch.no_grad():
                cat = lambda i, a: torch.cat([getattr(s[i], a).detach() for s in subs], 0)
                stack = lambda i, a: torch.stack([getattr(s[i], a).detach() for s in subs], 0)
                made = dict(w0=cat(0, 'weight'), b0=cat(0, 'bias'), **{k + str(i): stack(i, a) for i in (2, 4, 7) for k, a in (('w', 'weight'), ('b', 'bias'))})
            for k, t in made.items():
                if k[0] == 'w':
                    hip_ops.register_const_weight(t)   # their packed / transformed filters are made once, not per pass
            hit = (key, made)
            cache[stream] = hit
        return hit[1]

    def _windowed_tail_batched(self, frame0, frame1, crop, win, prep=None):
        ref = getattr(self, self._SUBNETS[0])
        sp = self._stacked_subnet_params()
        N = crop.size(0)
        # conv -> ReLU -> conv: the intermediate maps have one consumer each, which folds the producer's ReLU derivative into its data
        # gradient (the three element-wise passes over [4 N, 64, h, w] that the layers called one by one used to run: 2.5 ms of a C2 iteration)
        ch = self._chain(crop)
        c0, c2 = ({'want_defer': True} if ch else None), ({'want_defer': True} if ch else None)
        slope = lambda c: 0.0 if (c is not None and c.get('deferred')) else None
        x = ref[0](crop, params={'weight': sp['w0'], 'bias': sp['b0']}, act_slope=0.0, chain=c0)          # [N, 256, h, w]
        x = x.view(4 * N, 64, x.size(2), x.size(3))                                            # sample 4 n + s: Subnet s
        x = ref[2](x, params={'weight': sp['w2'], 'bias': sp['b2']}, act_slope=0.0, in_slope=slope(c0), chain=c2)
        c4 = {'want_defer': True} if ch else None
        x = ref[4](x, params={'weight': sp['w4'], 'bias': sp['b4']}, act_slope=0.0, in_slope=slope(c2), chain=c4)
        x = upsample_bilinear2x_window(x, win['half'], (win['crop'][0], win['crop'][2]), win['up'], True, slope(c4))
        # The taps leave the last convolution UNIT-MAJOR where the shapes allow (a sample [H][W / 16][51][16] instead of [51][H][W]): the
        # 51-tap op then reads a unit's taps as one contiguous run instead of 64-byte pieces of 51 planes.  The tensor keeps
        # its shape; only FunctionSepconvPair reads it.  SAVFI_SEPCONV_TAPS_PLANAR=1: the plain layout (A/B runs).
        height, width = win['up'][2] - 2, win['up'][3] - 2
        unit16 = (TAPS_UNIT16 and fuse_conv_act() and width % 16 == 0 and frames8_supported(frame0, N, 3, height, width, FILTER_TAPS, 4 * FILTER_TAPS)
                  and hip_ops.conv3x3_unit16_supported(x, sp['w7'], 0))
        # The tap GRADIENTS come back unit-major as well where only the convolution's data gradient will read them: the Subnets' own
        # weights carry no gradient in the inner loop (constant, detached); in the outer pass they do, and the weight-gradient and bias
        # kernels read the plain layout.  SAVFI_SEPCONV_GRADS_PLANAR=1: always the plain layout.
        grads16 = (unit16 and GRADS_UNIT16 and not sp['w7'].requires_grad and not sp['b7'].requires_grad
                   and hip_ops.conv3x3_in_unit16_supported((4 * N, FILTER_TAPS, height, width), sp['w7'], 0))
        if unit16:
            taps = hip_ops.conv_bias_act_tasks(x, sp['w7'], sp['b7'], ref[7].stride, 0, ref[7].dilation_rate, 1.0,
                                               getattr(ref[7], 'direct', False), None, False, 2 if grads16 else 1)
        else:
            taps = ref[7](x, params={'weight': sp['w7'], 'bias': sp['b7']}, padding=0)           # [4 N, 51, height, width]
        rim = (HALF,) * 4
        r0 = prep['rim0'] if prep is not None and 'rim0' in prep else F.pad(frame0, rim, mode='replicate')
        r1 = prep['rim1'] if prep is not None and 'rim1' in prep else F.pad(frame1, rim, mode='replicate')
        return FunctionSepconvPair.apply(r0, r1, taps, unit16, grads16)

    def _windowed_tail(self, frame0, frame1, combine, height, width, ph, pw, prep=None):
        win = self._window(height, width, ph, pw)
        cy0, cy1, cx0, cx1 = win['crop']
        crop = combine[:, :, cy0:cy1, cx0:cx1].contiguous()
        if (self.batch_subnets and not frame0.requires_grad and not frame1.requires_grad
                and FunctionSepconvPair.supported(frame0, crop.size(0), height, width, FILTER_TAPS)):
            return self._windowed_tail_batched(frame0, frame1, crop, win, prep)
        rim = (HALF,) * 4
        dot1 = FunctionSepconv.apply(F.pad(frame0, rim, mode='replicate'),
                                     self._subnet_window(self.moduleVertical1, crop, win),
                                     self._subnet_window(self.moduleHorizontal1, crop, win))
        dot2 = FunctionSepconv.apply(F.pad(frame1, rim, mode='replicate'),
                                     self._subnet_window(self.moduleVertical2, crop, win),
                                     self._subnet_window(self.moduleHorizontal2, crop, win))
        return dot1 + dot2

    def zero_grad(self, params=None):
        zero_grad_params(self, params)

    def restore_backup_stats(self):
        pass  # no batch statistics in this model
