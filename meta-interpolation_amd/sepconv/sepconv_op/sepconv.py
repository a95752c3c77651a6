"""FunctionSepconv -- drop-in for the reference op surface (sepconv/sepconv_op/sepconv.py:247-380).

    FunctionSepconv.apply(input[B,C,Ho+K-1,Wo+K-1], vertical[B,K,Ho,Wo], horizontal[B,K,Ho,Wo])
        -> output[B,C,Ho,Wo]
    backward(gradOutput) -> (gradInput | None, gradVertical | None, gradHorizontal | None)

Same argument meaning, same shape/contiguity asserts (reference :266-271, :314-317) and the same
NotImplementedError for CPU tensors (:293-294, :373-374).  The work is done by
savfi_sepconv_fwd_f32 / savfi_sepconv_bwd_f32 (include/savfi_hip.h) on torch's current stream;
there is no string templating, no JIT, no per-shape compile, and the outputs need no pre-zeroing.
"""
import torch

from ... import _hip


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
        ctx.save_for_backward(input, vertical, horizontal)
        output = torch.empty((B, C, Ho, Wo), dtype=input.dtype, device=input.device)
        lib = _hip.lib()
        _hip.launch("sepconv_fwd", lambda: _hip.check(lib.savfi_sepconv_fwd_f32(
            input.data_ptr(), vertical.data_ptr(), horizontal.data_ptr(), output.data_ptr(),
            B, C, Ho, Wo, K, _hip.current_stream()), "savfi_sepconv_fwd_f32"),
            nbytes=algorithmic_bytes(B, C, Ho, Wo, K))
        return output

    @staticmethod
    def backward(ctx, gradOutput):
        input, vertical, horizontal = ctx.saved_tensors
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
        if need_i or need_v or need_h:
            lib = _hip.lib()
            p = lambda t: None if t is None else t.data_ptr()
            name = "sepconv_bwd" if not need_i else "sepconv_bwd+gI"
            _hip.launch(name, lambda: _hip.check(lib.savfi_sepconv_bwd_f32(
                input.data_ptr(), vertical.data_ptr(), horizontal.data_ptr(), gradOutput.data_ptr(),
                p(gI), p(gV), p(gH), B, C, Ho, Wo, K, _hip.current_stream()), "savfi_sepconv_bwd_f32"),
                nbytes=algorithmic_bytes(B, C, Ho, Wo, K, grads=int(need_v) + int(need_h)))
        return gI, gV, gH


class ModuleSepconv(torch.nn.Module):
    """Module form kept for surface parity (reference :382-389); takes the three op inputs."""

    def forward(self, tensorInput, tensorVertical, tensorHorizontal):
        return FunctionSepconv.apply(tensorInput, tensorVertical, tensorHorizontal)
