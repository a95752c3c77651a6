"""FunctionSepconv -- drop-in for the reference op surface (sepconv/sepconv_op/sepconv.py:247-380).

    FunctionSepconv.apply(input[B,C,Ho+K-1,Wo+K-1], vertical[B,K,Ho,Wo], horizontal[B,K,Ho,Wo])
        -> output[B,C,Ho,Wo]
    backward(gradOutput) -> (gradInput | None, gradVertical | None, gradHorizontal | None)

Same argument meaning, same shape/contiguity asserts (reference :266-271, :314-317) and the same
NotImplementedError for CPU tensors (:293-294, :373-374).  The work is done by
savfi_sepconv_fwd_f32 / savfi_sepconv_bwd_f32 (include/savfi_hip.h) on torch's current stream;
there is no string templating, no JIT, no per-shape compile, and the outputs need no pre-zeroing.
"""
import os

import torch

from ... import _hip

# Frames of 8-bit images (include/savfi_hip.h, csrc/sepconv_ws.hip): the op classifies its frame tensor on the device at every forward
# call and hands the words to savfi_sepconv_{fwd,bwd}_frames8_f32, which run the three-product kernels on frames that are k / 255 and the
# six-product kernels on anything else.  SAVFI_SEPCONV_NO_FRAMES8=1: always the six-product kernels (A/B runs).
FRAMES8 = True            # A/B (module attributes): False = never the 8-bit-frame kernels
PAIR_ONE_LAUNCH = True    # False: the two local convolutions of the tail as two launches
FRAMES8_WORDS = 256
_E_UNSUPPORTED = -3


def frames8_supported(frame, B, C, Ho, Wo, K, tap_bstride=None):
    """shapes savfi_sepconv_*_frames8_f32 take (the wave-specialised kernels: K = 51, C = 3, Wo % 4 == 0, taps below 2^31 bytes)"""
    tb = K if tap_bstride is None else tap_bstride
    return (FRAMES8 and frame.is_cuda and frame.dtype == torch.float32 and K == 51 and C == 3 and Wo % 4 == 0
            and ((B - 1) * tb + K) * Ho * Wo * 4 < 2 ** 31
            and _hip.lib().savfi_sepconv_taps_strided_supported(B, C, Ho, Wo, K, tb) == 1)     # (the process's A/B switches included)


def frames8_classify(frame):
    """int32[FRAMES8_WORDS] on the frame's device: all zero = every element of `frame` is the fp32 quotient k / 255, k = 0..255"""
    cls = torch.empty(FRAMES8_WORDS, dtype=torch.int32, device=frame.device)
    lib = _hip.lib()
    _hip.launch("frames8_classify", lambda: _hip.check(lib.savfi_frames8_classify_f32(
        frame.data_ptr(), frame.numel(), cls.data_ptr(), _hip.current_stream()), "savfi_frames8_classify_f32"), nbytes=4 * frame.numel())
    return cls


def algorithmic_bytes(B, C, Ho, Wo, K, grads=0):
    """HBM bytes if every operand is read / written exactly once (fp32): input halo + v + h + out
    (+ gO is the `out`-sized term of the backward) + one [B,K,Ho,Wo] plane set per filter gradient."""
    return 4 * (B * C * (Ho + K - 1) * (Wo + K - 1) + 2 * B * K * Ho * Wo + B * C * Ho * Wo
                + grads * B * K * Ho * Wo)


def _dims(input, vertical, horizontal):
    B, C, Hi, Wi = input.shape
    K = min(vertical.size(1), horizontal.size(1))
    Ho = min(vertical.size(2), horizontal.size(2))
    Wo = min(vertical.size(3), horizontal.size(3))
    assert Hi - K == Ho - 1, "input height must be output height + K - 1"
    assert Wi - K == Wo - 1, "input width must be output width + K - 1"
    assert vertical.shape == horizontal.shape == (B, K, Ho, Wo), "vertical/horizontal must be [B,K,Ho,Wo]"
    return B, C, Ho, Wo, K


class FunctionSepconv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, vertical, horizontal):
        B, C, Ho, Wo, K = _dims(input, vertical, horizontal)
        assert input.is_contiguous() and vertical.is_contiguous() and horizontal.is_contiguous()
        if not input.is_cuda:
            raise NotImplementedError("FunctionSepconv has no CPU path (neither does the reference)")
        _hip.require_cuda(input, vertical, horizontal)
        output = torch.empty((B, C, Ho, Wo), dtype=input.dtype, device=input.device)
        lib = _hip.lib()
        cls = None
        if not ctx.needs_input_grad[0] and frames8_supported(input, B, C, Ho, Wo, K):
            cls = frames8_classify(input)
            rc = [0]

            def run8():
                rc[0] = lib.savfi_sepconv_fwd_frames8_f32(input.data_ptr(), vertical.data_ptr(), horizontal.data_ptr(), output.data_ptr(),
                                                          cls.data_ptr(), B, C, Ho, Wo, K, K, 0, _hip.current_stream())
                if rc[0] != _E_UNSUPPORTED:
                    _hip.check(rc[0], "savfi_sepconv_fwd_frames8_f32")
            _hip.launch("sepconv_fwd", run8, nbytes=algorithmic_bytes(B, C, Ho, Wo, K))
            if rc[0] == _E_UNSUPPORTED:
                cls = None
        if cls is None:
            _hip.launch("sepconv_fwd", lambda: _hip.check(lib.savfi_sepconv_fwd_f32(
                input.data_ptr(), vertical.data_ptr(), horizontal.data_ptr(), output.data_ptr(),
                B, C, Ho, Wo, K, _hip.current_stream()), "savfi_sepconv_fwd_f32"),
                nbytes=algorithmic_bytes(B, C, Ho, Wo, K))
            ctx.save_for_backward(input, vertical, horizontal)
        else:
            ctx.save_for_backward(input, vertical, horizontal, cls)
        return output

    @staticmethod
    def backward(ctx, gradOutput):
        input, vertical, horizontal = ctx.saved_tensors[:3]
        cls = ctx.saved_tensors[3] if len(ctx.saved_tensors) > 3 else None
        B, C, Ho, Wo, K = _dims(input, vertical, horizontal)
        if not gradOutput.is_contiguous():
            gradOutput = gradOutput.contiguous()
        if not input.is_cuda:
            raise NotImplementedError("FunctionSepconv has no CPU path (neither does the reference)")
        _hip.require_cuda(gradOutput)
        need_i, need_v, need_h = ctx.needs_input_grad
        gI = torch.empty_like(input) if need_i else None
        gV = torch.empty_like(vertical) if need_v else None
        gH = torch.empty_like(horizontal) if need_h else None
        if cls is not None and need_v and need_h and not need_i:
            lib = _hip.lib()
            _hip.launch("sepconv_bwd", lambda: _hip.check(lib.savfi_sepconv_bwd_frames8_f32(
                input.data_ptr(), vertical.data_ptr(), horizontal.data_ptr(), gradOutput.data_ptr(), gV.data_ptr(), gH.data_ptr(),
                cls.data_ptr(), B, C, Ho, Wo, K, K, 0, _hip.current_stream()), "savfi_sepconv_bwd_frames8_f32"),
                nbytes=algorithmic_bytes(B, C, Ho, Wo, K, grads=2))
        elif need_i or need_v or need_h:
            lib = _hip.lib()
            p = lambda t: None if t is None else t.data_ptr()
            name = "sepconv_bwd" if not need_i else "sepconv_bwd+gI"
            _hip.launch(name, lambda: _hip.check(lib.savfi_sepconv_bwd_f32(
                input.data_ptr(), vertical.data_ptr(), horizontal.data_ptr(), gradOutput.data_ptr(),
                p(gI), p(gV), p(gH), B, C, Ho, Wo, K, _hip.current_stream()), "savfi_sepconv_bwd_f32"),
                nbytes=algorithmic_bytes(B, C, Ho, Wo, K, grads=int(need_v) + int(need_h)))
        return gI, gV, gH


class FunctionSepconvPair(torch.autograd.Function):
    """sepconv(input0, taps[0::4], taps[1::4]) + sepconv(input1, taps[2::4], taps[3::4]) on an INTERLEAVED tap tensor.

        FunctionSepconvPair.apply(input0[B,C,Ho+K-1,Wo+K-1], input1[...], taps[4*B,K,Ho,Wo]) -> output[B,C,Ho,Wo]

    taps[4 b + s] are the K tap planes of sample b from sub-network s (0: vertical1, 1: horizontal1, 2: vertical2, 3: horizontal2) --
    the layout the plugin's four Subnets leave when they run as ONE task-batched launch per layer (sepconv/model.py).  The two local
    convolutions of the reference (sepconv/model.py:346-347 -> sepconv_op/sepconv.py:247-380) read that buffer in place and their filter
    gradients are written into one buffer of the same layout (savfi_sepconv_*_taps_strided_f32, tap_bstride = 4 K): no slices are
    copied out, no gradients are concatenated.  The frames carry no gradient on this path (needs_input_grad is asserted).

    taps_unit16=True: the MEMORY of `taps` is unit-major -- sample [Ho][Wo / 16][K][16] instead of [K][Ho][Wo], what the plugin's last Subnet
    convolution writes through hip_ops.conv_bias_act_tasks(..., out_unit16=True) -- and the kernels read a unit's 51 taps x 16 pixels as one
    contiguous run (include/savfi_hip.h `taps_unit16`; frames8 entry points only: check frames8_supported and Wo % 16 == 0 first).  The
    returned tap gradient is laid out as its shape says -- unless grads_unit16=True: then its memory is unit-major too, for a producer
    whose backward reads that (hip_ops.conv_bias_act_tasks(..., out_unit16=2): data gradient only)."""

    @staticmethod
    def supported(frame, batch, height, width, taps=51):
        """Can the strided entry points take `batch` samples of a [*, 3, height, width] frame (savfi_sepconv_*_taps_strided_f32: the
        wave-specialised kernels -- K = 51, C = 3, width % 4 == 0, the interleaved tap tensor below 2^31 bytes)?"""
        return (frame.is_cuda and frame.dtype == torch.float32 and frame.size(1) == 3 and taps == 51 and width % 4 == 0
                and 4 * batch * taps * height * width * 4 < 2 ** 31
                # ... and the library takes it in this process: the A/B switches (SAVFI_SEPCONV_NO_WS, _NO_MFMA, ...) make the strided
                # entry points refuse, and this op has no other path -- the caller then runs the Subnets one by one
                and _hip.lib().savfi_sepconv_taps_strided_supported(batch, 3, height, width, taps, 4 * taps) == 1)

    @staticmethod
    def forward(ctx, input0, input1, taps, taps_unit16=False, grads_unit16=False):
        B, C, Hi, Wi = input0.shape
        K, Ho, Wo = taps.shape[1:]
        assert input1.shape == input0.shape and taps.size(0) == 4 * B and Hi - K == Ho - 1 and Wi - K == Wo - 1, (input0.shape, taps.shape)
        assert input0.is_contiguous() and input1.is_contiguous() and taps.is_contiguous()
        from ... import hip_ops
        hip_ops.require_layout(taps, hip_ops.UNIT16 if taps_unit16 else None, "FunctionSepconvPair(taps_unit16=%s): taps" % bool(taps_unit16))
        _hip.require_cuda(input0, input1, taps)
        lib, st = _hip.lib(), _hip.current_stream()
        plane = K * Ho * Wo * 4
        words = [frames8_classify(inp) for inp in (input0, input1)] if frames8_supported(input0, B, C, Ho, Wo, K, 4 * K) else None
        u16 = 1 if taps_unit16 else 0
        assert not u16 or (words is not None and Wo % 16 == 0), "unit-major taps: the frames8 entry points, widths that are a multiple of 16"
        assert not grads_unit16 or u16, "unit-major gradients go with unit-major taps"
        ctx.taps_unit16 = u16 | (2 if grads_unit16 else 0)
        # the pair launch holds 2 B virtual samples: its own limit (a very large B on tiny frames falls back to two launches)
        ctx.pair = bool(words is not None and PAIR_ONE_LAUNCH and frames8_supported(input0, 2 * B, C, Ho, Wo, K, 2 * K))      # (virtual sample 2 b + f: taps 2 K planes apart)
        if ctx.pair:
            # ONE launch for both local convolutions (2 B virtual samples; see backward): out2[b, f] = the convolution of frame f
            out2 = torch.empty((B, 2, C, Ho, Wo), dtype=input0.dtype, device=input0.device)
            _hip.launch("sepconv_fwd", lambda: _hip.check(lib.savfi_sepconv_fwd_pair_frames8_f32(
                input0.data_ptr(), input1.data_ptr(), taps.data_ptr(), out2.data_ptr(), words[0].data_ptr(), words[1].data_ptr(),
                B, C, Ho, Wo, K, u16, st), "savfi_sepconv_fwd_pair_frames8_f32"), nbytes=2 * algorithmic_bytes(B, C, Ho, Wo, K))
            ctx.save_for_backward(input0, input1, taps, *words)
            return out2[:, 0].add(out2[:, 1])
        out0 = torch.empty((B, C, Ho, Wo), dtype=input0.dtype, device=input0.device)
        out1 = torch.empty_like(out0)
        for i, (inp, out, s) in enumerate(((input0, out0, 0), (input1, out1, 2))):
            if words is not None:
                _hip.launch("sepconv_fwd", lambda inp=inp, out=out, s=s, i=i: _hip.check(lib.savfi_sepconv_fwd_frames8_f32(
                    inp.data_ptr(), taps.data_ptr() + s * plane, taps.data_ptr() + (s + 1) * plane, out.data_ptr(), words[i].data_ptr(),
                    B, C, Ho, Wo, K, 4 * K, u16, st), "savfi_sepconv_fwd_frames8_f32"), nbytes=algorithmic_bytes(B, C, Ho, Wo, K))
            else:
                _hip.launch("sepconv_fwd", lambda inp=inp, out=out, s=s: _hip.check(lib.savfi_sepconv_fwd_taps_strided_f32(
                    inp.data_ptr(), taps.data_ptr() + s * plane, taps.data_ptr() + (s + 1) * plane, out.data_ptr(),
                    B, C, Ho, Wo, K, 4 * K, st), "savfi_sepconv_fwd_taps_strided_f32"), nbytes=algorithmic_bytes(B, C, Ho, Wo, K))
        ctx.save_for_backward(input0, input1, taps, *(words or ()))
        return out0.add_(out1)

    @staticmethod
    def backward(ctx, gradOutput):
        input0, input1, taps = ctx.saved_tensors[:3]
        words = ctx.saved_tensors[3:] or None
        assert not ctx.needs_input_grad[0] and not ctx.needs_input_grad[1], "FunctionSepconvPair: frames carry no gradient on this path"
        if not ctx.needs_input_grad[2]:
            return None, None, None, None, None
        B, C = input0.shape[:2]
        K, Ho, Wo = taps.shape[1:]
        gradOutput = gradOutput.contiguous()
        _hip.require_cuda(gradOutput)
        from ... import hip_ops
        gT = torch.empty_like(taps)
        lib, st = _hip.lib(), _hip.current_stream()
        plane = K * Ho * Wo * 4
        if ctx.pair:
            # ONE launch for both local convolutions (2 B virtual samples): a launch of these kernels has a fixed cost of ~24 us
            _hip.launch("sepconv_bwd", lambda: _hip.check(lib.savfi_sepconv_bwd_pair_frames8_f32(
                input0.data_ptr(), input1.data_ptr(), taps.data_ptr(), gradOutput.data_ptr(), gT.data_ptr(), words[0].data_ptr(),
                words[1].data_ptr(), B, C, Ho, Wo, K, ctx.taps_unit16, st), "savfi_sepconv_bwd_pair_frames8_f32"),
                nbytes=2 * algorithmic_bytes(B, C, Ho, Wo, K, grads=2))
            if ctx.taps_unit16 & 2:
                hip_ops.tag_layout(gT, hip_ops.UNIT16)
            return None, None, gT, None, None
        for i, (inp, s) in enumerate(((input0, 0), (input1, 2))):
            if words is not None:
                _hip.launch("sepconv_bwd", lambda inp=inp, s=s, i=i: _hip.check(lib.savfi_sepconv_bwd_frames8_f32(
                    inp.data_ptr(), taps.data_ptr() + s * plane, taps.data_ptr() + (s + 1) * plane, gradOutput.data_ptr(),
                    gT.data_ptr() + s * plane, gT.data_ptr() + (s + 1) * plane, words[i].data_ptr(), B, C, Ho, Wo, K, 4 * K, ctx.taps_unit16, st),
                    "savfi_sepconv_bwd_frames8_f32"), nbytes=algorithmic_bytes(B, C, Ho, Wo, K, grads=2))
            else:
                _hip.launch("sepconv_bwd", lambda inp=inp, s=s: _hip.check(lib.savfi_sepconv_bwd_taps_strided_f32(
                    inp.data_ptr(), taps.data_ptr() + s * plane, taps.data_ptr() + (s + 1) * plane, gradOutput.data_ptr(),
                    gT.data_ptr() + s * plane, gT.data_ptr() + (s + 1) * plane, B, C, Ho, Wo, K, 4 * K, st),
                    "savfi_sepconv_bwd_taps_strided_f32"), nbytes=algorithmic_bytes(B, C, Ho, Wo, K, grads=2))
        if ctx.taps_unit16 & 2:
            hip_ops.tag_layout(gT, hip_ops.UNIT16)
        return None, None, gT, None, None


class ModuleSepconv(torch.nn.Module):
    """Module form kept for surface parity (reference :382-389); takes the three op inputs."""

    def forward(self, tensorInput, tensorVertical, tensorHorizontal):
        return FunctionSepconv.apply(tensorInput, tensorVertical, tensorHorizontal)
