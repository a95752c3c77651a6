"""-m gpu: the SepConv op on frames of 8-bit images (savfi_frames8_classify_f32, savfi_sepconv_{fwd,bwd}_frames8_f32; csrc/sepconv_ws.hip).

The reference feeds the op decoded PNG frames, k / 255 (data/vimeo_septuplet.py:24-36, sepconv/model.py:346-347).  For such a frame tensor the
device selects kernels that hold the window as the integers k (one exact bf16 piece) and spend three bf16 products per fp32 product; for any
other tensor it selects the six-product kernels.  Checked here: the classifier, both selections through the C ABI against the CPU oracle at
the tolerance of tests/test_hip_ops_gpu.py (1e-5 of max|ref|: fp32 sums of 2601 products in another association), that the selection is
what the words say, and that tensors which do not qualify get bit-identical results to the entry points without the words."""
import math

import pytest
import torch

from meta_interpolation_amd import _hip
from meta_interpolation_amd.sepconv.sepconv_op import sepconv as S
from oracle import torch_ops as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
K = 51


def _rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


def _inputs(B, Ho, Wo, seed, frames8=True):
    g = torch.Generator().manual_seed(seed)
    if frames8:
        inp = torch.randint(0, 256, (B, 3, Ho + K - 1, Wo + K - 1), generator=g).float().div(255)      # ToTensor's arithmetic
    else:
        inp = torch.rand(B, 3, Ho + K - 1, Wo + K - 1, generator=g)
    v = torch.randn(B, K, Ho, Wo, generator=g) / math.sqrt(K)
    h = torch.randn(B, K, Ho, Wo, generator=g) / math.sqrt(K)
    gO = torch.randn(B, 3, Ho, Wo, generator=g)
    return inp, v, h, gO


def _words(x):
    w = S.frames8_classify(x)
    torch.cuda.synchronize()
    return w


def test_classifier_accepts_k_over_255_and_nothing_else():
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, 256, (2, 3, 77, 131), generator=g).float().div(255).to(DEV)
    assert int(_words(x).abs().sum()) == 0
    # float64 quotient rounded to fp32 (numpy's route in synthetic.py) is the same number or its neighbour: accepted
    x64 = (torch.randint(0, 256, (5, 3, 33, 47), generator=g).double() / 255.0).float().to(DEV)
    assert int(_words(x64).abs().sum()) == 0
    for bad in (0.5, 1.0 + 1e-6, -1.0 / 255, 3e-5, float("nan"), float("inf"), 100.0 / 255 * (1 + 1e-6)):
        y = x.clone()
        y.view(-1)[12345] = bad
        assert int(_words(y).abs().sum()) > 0, bad
    # one odd element at either end of a tensor whose length is no multiple of 4, and a misaligned base pointer
    z = torch.randint(0, 256, (4099,), generator=g).float().div(255).to(DEV)
    assert int(_words(z).abs().sum()) == 0
    assert int(_words(z[1:]).abs().sum()) == 0
    for pos in (0, 4098):
        y = z.clone()
        y[pos] = 0.3
        assert int(_words(y).abs().sum()) > 0
    assert int(_words(torch.rand(1, 3, 64, 64, device=DEV)).abs().sum()) > 0


@pytest.mark.parametrize("B,Ho,Wo", [(1, 16, 32), (2, 37, 36), (1, 128, 128), (2, 64, 96), (1, 9, 68), (3, 5, 4)])
def test_frames8_kernels_vs_oracle(B, Ho, Wo):
    inp, v, h, gO = _inputs(B, Ho, Wo, seed=100 * B + Ho)
    ref = O.sepconv_forward_c(inp, v, h)
    _, rV, rH = O.sepconv_backward_c(inp, v, h, gO)
    di = inp.to(DEV)
    dv, dh = (t.to(DEV).requires_grad_() for t in (v, h))
    assert S.frames8_supported(di, B, 3, Ho, Wo, K)
    out = S.FunctionSepconv.apply(di, dv, dh)
    out.backward(gO.to(DEV))
    torch.cuda.synchronize()
    assert _rel(out.detach().cpu(), ref) < 1e-5
    assert _rel(dv.grad.cpu(), rV) < 1e-5
    assert _rel(dh.grad.cpu(), rH) < 1e-5
    assert _hip.lib().savfi_sepconv_ws_errors() == 0


def _abi_fwd(inp, v, h, words, tb=K, u16=0):
    B, _, Ho, Wo = v.shape[0], None, v.shape[2], v.shape[3]
    out = torch.empty(B, 3, Ho, Wo, device=DEV)
    lib = _hip.lib()
    if words is None:
        _hip.check(lib.savfi_sepconv_fwd_taps_strided_f32(inp.data_ptr(), v.data_ptr(), h.data_ptr(), out.data_ptr(), B, 3, Ho, Wo, K, tb,
                                                          _hip.current_stream()), "fwd")
    else:
        _hip.check(lib.savfi_sepconv_fwd_frames8_f32(inp.data_ptr(), v.data_ptr(), h.data_ptr(), out.data_ptr(), words.data_ptr(), B, 3, Ho, Wo,
                                                     K, tb, u16, _hip.current_stream()), "fwd8")
    return out


def _abi_bwd(inp, v, h, gO, words, tb=K, u16=0):
    B, Ho, Wo = v.shape[0], v.shape[2], v.shape[3]
    gV, gH = torch.empty_like(v), torch.empty_like(h)
    lib = _hip.lib()
    if words is None:
        _hip.check(lib.savfi_sepconv_bwd_taps_strided_f32(inp.data_ptr(), v.data_ptr(), h.data_ptr(), gO.data_ptr(), gV.data_ptr(), gH.data_ptr(),
                                                          B, 3, Ho, Wo, K, tb, _hip.current_stream()), "bwd")
    else:
        _hip.check(lib.savfi_sepconv_bwd_frames8_f32(inp.data_ptr(), v.data_ptr(), h.data_ptr(), gO.data_ptr(), gV.data_ptr(), gH.data_ptr(),
                                                     words.data_ptr(), B, 3, Ho, Wo, K, tb, u16, _hip.current_stream()), "bwd8")
    return gV, gH


def test_the_words_select_the_kernel_on_the_device():
    B, Ho, Wo = 2, 40, 64
    zero = torch.zeros(S.FRAMES8_WORDS, dtype=torch.int32, device=DEV)
    one = torch.zeros(S.FRAMES8_WORDS, dtype=torch.int32, device=DEV)
    one[S.FRAMES8_WORDS - 1] = 1
    # (a) frames that do not qualify + words that say so: bit-identical to the entry points without the words
    inp, v, h, gO = (t.to(DEV) for t in _inputs(B, Ho, Wo, seed=5, frames8=False))
    words = _words(inp)
    assert int(words.abs().sum()) > 0
    assert torch.equal(_abi_fwd(inp, v, h, words), _abi_fwd(inp, v, h, None))
    for a, b in zip(_abi_bwd(inp, v, h, gO, words), _abi_bwd(inp, v, h, gO, None)):
        assert torch.equal(a, b)
    # (b) the same frames with forged all-zero words: the three-product kernels run (they round 255 w to an integer: visibly wrong here)
    assert _rel(_abi_fwd(inp, v, h, zero), _abi_fwd(inp, v, h, None)) > 1e-4
    assert _rel(_abi_bwd(inp, v, h, gO, zero)[0], _abi_bwd(inp, v, h, gO, None)[0]) > 1e-4
    # (c) frames that qualify with forged non-zero words: the six-product kernels, bit for bit; with their own words: the same to fp32 rounding
    inp8 = _inputs(B, Ho, Wo, seed=6)[0].to(DEV)
    assert torch.equal(_abi_fwd(inp8, v, h, one), _abi_fwd(inp8, v, h, None))
    w8 = _words(inp8)
    assert int(w8.abs().sum()) == 0
    assert _rel(_abi_fwd(inp8, v, h, w8), _abi_fwd(inp8, v, h, None)) < 2e-6
    for a, b in zip(_abi_bwd(inp8, v, h, gO, w8), _abi_bwd(inp8, v, h, gO, None)):
        assert _rel(a, b) < 2e-6
    assert _hip.lib().savfi_sepconv_ws_errors() == 0


def test_frames8_is_as_close_to_float64_as_the_six_product_kernels():
    """against a float64 evaluation of the op and its autograd on the fp32 inputs (oracle/torch_ops.sepconv_torch in double)"""
    B, Ho, Wo = 1, 32, 64
    inp, v, h, gO = _inputs(B, Ho, Wo, seed=11)
    v64, h64 = v.double().requires_grad_(), h.double().requires_grad_()
    ref = O.sepconv_torch(inp.double(), v64, h64)
    ref.backward(gO.double())
    d = [t.to(DEV) for t in (inp, v, h)]
    w8 = _words(d[0])
    err = lambda x, r: (x.cpu().double() - r).abs().max().item() / r.abs().max().item()
    e8, e6 = err(_abi_fwd(*d, w8), ref.detach()), err(_abi_fwd(*d, None), ref.detach())
    g8, g6 = _abi_bwd(*d, gO.to(DEV), w8), _abi_bwd(*d, gO.to(DEV), None)
    print("max error / max|ref| against float64: forward %.3g (three products) %.3g (six); gV %.3g %.3g; gH %.3g %.3g"
          % (e8, e6, err(g8[0], v64.grad), err(g6[0], v64.grad), err(g8[1], h64.grad), err(g6[1], h64.grad)))
    assert e8 < max(2 * e6, 5e-7)
    assert err(g8[0], v64.grad) < max(2 * err(g6[0], v64.grad), 5e-7)
    assert err(g8[1], h64.grad) < max(2 * err(g6[1], h64.grad), 5e-7)


def test_full_size_frames8_against_the_six_product_kernels_and_refusals():
    B, Ho, Wo = 4, 256, 448
    inp, v, h, gO = (t.to(DEV) for t in _inputs(B, Ho, Wo, seed=21))
    w8 = _words(inp)
    assert int(w8.abs().sum()) == 0
    assert _rel(_abi_fwd(inp, v, h, w8), _abi_fwd(inp, v, h, None)) < 2e-6
    for a, b in zip(_abi_bwd(inp, v, h, gO, w8), _abi_bwd(inp, v, h, gO, None)):
        assert _rel(a, b) < 2e-6
    assert _hip.lib().savfi_sepconv_ws_errors() == 0
    lib = _hip.lib()
    out = torch.empty(B, 3, Ho, Wo, device=DEV)
    args = (inp.data_ptr(), v.data_ptr(), h.data_ptr(), out.data_ptr())
    assert lib.savfi_sepconv_fwd_frames8_f32(*args, None, B, 3, Ho, Wo, K, K, 0, _hip.current_stream()) == -1          # NULL words
    assert lib.savfi_sepconv_fwd_frames8_f32(*args, w8.data_ptr(), B, 3, Ho, Wo - 2, K, K, 0, _hip.current_stream()) == -3   # width % 4
    assert lib.savfi_sepconv_fwd_frames8_f32(*args, w8.data_ptr(), B, 3, Ho, Wo, 13, 13, 0, _hip.current_stream()) == -3     # K != 51
    assert lib.savfi_sepconv_fwd_frames8_f32(*args, w8.data_ptr(), B, 3, Ho, Wo - 4, K, K, 1, _hip.current_stream()) == -3   # unit-major taps: width % 16
    assert lib.savfi_frames8_classify_f32(None, 5, w8.data_ptr(), _hip.current_stream()) == -1
    assert lib.savfi_frames8_classify_f32(inp.data_ptr(), 0, w8.data_ptr(), _hip.current_stream()) == -2


def test_op_falls_back_for_shapes_and_tensors_outside_the_fast_path(monkeypatch):
    # width % 4 != 0, a frame that carries a gradient, the switch: all through the entry points without the words, same numbers as before
    inp, v, h, gO = _inputs(1, 20, 30, seed=8)
    di, dv, dh = inp.to(DEV), v.to(DEV).requires_grad_(), h.to(DEV).requires_grad_()
    assert not S.frames8_supported(di, 1, 3, 20, 30, K)
    out = S.FunctionSepconv.apply(di, dv, dh)
    out.backward(gO.to(DEV))
    assert _rel(out.detach().cpu(), O.sepconv_forward_c(inp, v, h)) < 1e-5
    inp, v, h, gO = _inputs(1, 24, 40, seed=9)
    di, dv, dh = (t.to(DEV).requires_grad_() for t in (inp, v, h))
    out = S.FunctionSepconv.apply(di, dv, dh)
    out.backward(gO.to(DEV))
    rI, rV, rH = O.sepconv_backward_c(inp, v, h, gO, need_input=True)
    assert _rel(di.grad.cpu(), rI) < 1e-5 and _rel(dv.grad.cpu(), rV) < 1e-5 and _rel(dh.grad.cpu(), rH) < 1e-5
    monkeypatch.setattr(S, "FRAMES8", False)
    d2 = inp.to(DEV)
    dv2, dh2 = v.to(DEV).requires_grad_(), h.to(DEV).requires_grad_()
    out2 = S.FunctionSepconv.apply(d2, dv2, dh2)
    out2.backward(gO.to(DEV))
    monkeypatch.setattr(S, "FRAMES8", True)
    dv3, dh3 = v.to(DEV).requires_grad_(), h.to(DEV).requires_grad_()
    out3 = S.FunctionSepconv.apply(d2, dv3, dh3)
    out3.backward(gO.to(DEV))
    assert _rel(out3.detach(), out2.detach()) < 2e-6 and _rel(dv3.grad, dv2.grad) < 2e-6 and _rel(dh3.grad, dh2.grad) < 2e-6


def test_pair_op_on_interleaved_taps_with_frames8(monkeypatch):
    B, Ho, Wo = 2, 36, 64
    g = torch.Generator().manual_seed(31)
    f0 = torch.randint(0, 256, (B, 3, Ho + K - 1, Wo + K - 1), generator=g).float().div(255).to(DEV)
    f1 = torch.randint(0, 256, (B, 3, Ho + K - 1, Wo + K - 1), generator=g).float().div(255).to(DEV)
    taps = (torch.randn(4 * B, K, Ho, Wo, generator=g) / math.sqrt(K)).to(DEV)
    gO = torch.randn(B, 3, Ho, Wo, generator=g).to(DEV)
    res = []
    for on in (True, False):
        monkeypatch.setattr(S, "FRAMES8", on)
        t = taps.clone().requires_grad_()
        out = S.FunctionSepconvPair.apply(f0, f1, t)
        out.backward(gO)
        res.append((out.detach(), t.grad))
    assert _rel(res[0][0], res[1][0]) < 2e-6 and _rel(res[0][1], res[1][1]) < 2e-6
    ref = O.sepconv_forward_c(f0.cpu(), taps[0::4].cpu().contiguous(), taps[1::4].cpu().contiguous()) + \
        O.sepconv_forward_c(f1.cpu(), taps[2::4].cpu().contiguous(), taps[3::4].cpu().contiguous())
    assert _rel(res[0][0].cpu(), ref) < 1e-5
    assert _hip.lib().savfi_sepconv_ws_errors() == 0


def _to_unit16(t):
    """[B,K,H,W] values -> a tensor of the same shape whose MEMORY is [B][H][W/16][K][16] (what savfi_conv3x3_tasks_pre_unit16_f32 writes)"""
    B, Kk, H, W = t.shape
    return t.view(B, Kk, H, W // 16, 16).permute(0, 2, 3, 1, 4).contiguous().view(B, Kk, H, W)


@pytest.mark.parametrize("B,Ho,Wo,f8", [(2, 37, 48, True), (1, 64, 96, True), (3, 5, 16, True), (2, 40, 64, False)])
def test_unit_major_taps_give_the_same_bits(B, Ho, Wo, f8):
    """taps_unit16: the same values read from the unit-major layout -- identical results, forward and both filter gradients, on the
    three-product kernels and (frames that do not qualify) on the six-product ones; the gradients come back [K][Ho][Wo]"""
    inp, v, h, gO = (t.to(DEV) for t in _inputs(B, Ho, Wo, seed=77 + Ho, frames8=f8))
    words = _words(inp)
    assert (int(words.abs().sum()) == 0) == f8
    vu, hu = _to_unit16(v), _to_unit16(h)
    assert torch.equal(_abi_fwd(inp, vu, hu, words, u16=1), _abi_fwd(inp, v, h, words))
    for a, b in zip(_abi_bwd(inp, vu, hu, gO, words, u16=1), _abi_bwd(inp, v, h, gO, words)):
        assert torch.equal(a, b)
    assert _hip.lib().savfi_sepconv_ws_errors() == 0


def test_winograd_convolution_writes_unit_major_and_the_pair_op_reads_it():
    """hip_ops.conv_bias_act_tasks(out_unit16=True) -> FunctionSepconvPair(taps_unit16=True) against the plain layout: the convolution's
    result rearranged is bit-identical, the op's output and every gradient (taps -> convolution input, weights, bias) are bit-identical"""
    from meta_interpolation_amd import hip_ops
    B, T, C, Ho, Wo = 2, 4, K, 96, 128
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B * T, C, Ho + 2, Wo + 2, generator=g).to(DEV)
    w = (torch.randn(T, K, C, 3, 3, generator=g) / (3 * math.sqrt(C))).to(DEV)
    b = (torch.randn(T, K, generator=g) * 0.1).to(DEV)
    f0 = torch.randint(0, 256, (B, 3, Ho + K - 1, Wo + K - 1), generator=g).float().div(255).to(DEV)
    f1 = torch.randint(0, 256, (B, 3, Ho + K - 1, Wo + K - 1), generator=g).float().div(255).to(DEV)
    gO = torch.randn(B, 3, Ho, Wo, generator=g).to(DEV)
    assert hip_ops.conv3x3_unit16_supported(x, w, 0)
    # (a launch too small for the F(4x4) kernel to pay -- 40 workgroups -- runs the F(2x2) form, kind 'wino2', whose reduction is split: the
    # plugin's gate says no; the library's form-0 entry point itself would take it:
    # test_unit_major_convolution_entry_points_refuse_what_they_cannot_do)
    assert not hip_ops.conv3x3_unit16_supported(x[:, :, :38, :66].contiguous(), w, 0)
    res = []
    for u16 in (False, True):
        xs, ws, bs = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
        taps = hip_ops.conv_bias_act_tasks(xs, ws, bs, 1, 0, 1, 1.0, False, None, False, u16)
        out = S.FunctionSepconvPair.apply(f0, f1, taps, u16)
        out.backward(gO)
        res.append((taps.detach(), out.detach(), xs.grad, ws.grad, bs.grad))
    assert torch.equal(res[1][0], _to_unit16(res[0][0]))
    for a, bb in zip(res[0][1:], res[1][1:]):
        assert torch.equal(a, bb)
    # the tap gradients unit-major as well (constant weights: the convolution's backward is its data gradient only, which reads them so)
    assert hip_ops.conv3x3_in_unit16_supported((B * T, K, Ho, Wo), w, 0)
    xs = x.clone().requires_grad_()
    taps = hip_ops.conv_bias_act_tasks(xs, w, b, 1, 0, 1, 1.0, False, None, False, 2)
    out = S.FunctionSepconvPair.apply(f0, f1, taps, True, True)
    out.backward(gO)
    assert torch.equal(out.detach(), res[0][1]) and torch.equal(xs.grad, res[0][2])
    # ... and the op alone: gV / gH unit-major are the planar ones rearranged
    inp, v, h, g2 = (t.to(DEV) for t in _inputs(2, 40, 64, seed=3))
    w8 = _words(inp)
    for a, bb in zip(_abi_bwd(inp, _to_unit16(v), _to_unit16(h), g2, w8, u16=3), _abi_bwd(inp, v, h, g2, w8)):
        assert torch.equal(a, _to_unit16(bb))
    assert _hip.lib().savfi_sepconv_ws_errors() == 0


def test_unit_major_convolution_entry_points_refuse_what_they_cannot_do():
    from meta_interpolation_amd import hip_ops
    lib, st = _hip.lib(), _hip.current_stream()
    N, T, C, H, W = 8, 4, K, 98, 130
    x = torch.randn(N, C, H, W, device=DEV)
    u = hip_ops.conv3x3_filters(torch.randn(T, C, C, 3, 3, device=DEV), True, True)
    out = torch.empty(N, C, H - 2, W - 2, device=DEV)
    P = lambda t: t.data_ptr()
    assert lib.savfi_conv3x3_unit16_supported(N, T, C, C, H, W, 0) == 1
    assert lib.savfi_conv3x3_tasks_pre_unit16_f32(None, P(u[0]), None, P(out), N, T, C, C, H, W, 0, 1.0, st) == -1
    assert lib.savfi_conv3x3_tasks_pre_unit16_f32(P(x), P(u[0]), None, P(out), N, T, C, C, H, W - 2, 0, 1.0, st) == -3      # width 126: not % 16
    # a deep layer on a small map: the reduction is split over workgroups (F(4x4) from 256 channels, F(2x2) beyond 512) -> refused
    C2 = 320
    u2 = hip_ops.conv3x3_filters(torch.randn(T, C2, C2, 3, 3, device=DEV), True, True)
    x2, out2 = torch.randn(4, C2, 14, 18, device=DEV), torch.empty(4, C2, 12, 16, device=DEV)       # 40 workgroups: split
    assert lib.savfi_conv3x3_f4_workgroups(4, 640, 640, 14, 18, 0, 0) == 0 and lib.savfi_conv3x3_f4_workgroups(4, C, C, 38, 66, 0, 0) == 4 * 5 * 2
    assert lib.savfi_conv3x3_tasks_pre_workspace_floats(4, T, C2, C2, 14, 18, 0, 0) > 0
    assert lib.savfi_conv3x3_unit16_supported(4, T, C2, C2, 14, 18, 0) == 0
    assert lib.savfi_conv3x3_tasks_pre_unit16_f32(P(x2), P(u2[0]), None, P(out2), 4, T, C2, C2, 14, 18, 0, 1.0, st) == -3
    assert lib.savfi_conv3x3_tasks_pre_unit16_f32(P(x), P(u[0]), None, P(out), 4, T, C, C, 38, 66, 0, 1.0, st) == 0         # F(4x4): no split
    assert lib.savfi_conv3x3_unit16_supported(N, T, C, C, H, W - 2, 0) == 0 and lib.savfi_conv3x3_unit16_supported(N, 3, C, C, H, W, 0) == 0
    gy = torch.randn(N, C, H - 2, W - 2, device=DEV)
    assert lib.savfi_conv3x3_in_unit16_supported(N, T, C, C, H - 2, W - 2, 0) == 1
    assert lib.savfi_conv3x3_dgrad_in_unit16_f32(P(gy), None, P(x), N, T, C, C, H - 2, W - 2, 0, st) == -1
    assert lib.savfi_conv3x3_dgrad_in_unit16_f32(P(gy), P(u[1]), P(x), N, T, C, C, H - 2, W - 4, 0, st) == -3              # width 126
    assert lib.savfi_conv3x3_in_unit16_supported(N, T, C, C, H - 2, W - 4, 0) == 0
    # the Python gate of the plugin says no where the direct split-bf16 kernel would run the layer (64 -> 64 channels on a launch too small
    # to fill the chip with F(4x4) workgroups)
    assert not hip_ops.conv3x3_unit16_supported(torch.randn(4, 64, 38, 66, device=DEV), torch.randn(4, 64, 64, 3, 3, device=DEV), 0)
    assert hip_ops.conv3x3_unit16_supported(torch.randn(8, 64, 98, 130, device=DEV), torch.randn(4, 64, 64, 3, 3, device=DEV), 0)
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,Ho,Wo,f8,u16", [(2, 37, 36, True, 0), (3, 40, 64, True, 1), (2, 64, 96, True, 3), (1, 24, 48, False, 0), (2, 33, 32, "one", 1)])
def test_pair_launch_gives_the_bits_of_the_two_launches(B, Ho, Wo, f8, u16):
    """savfi_sepconv_bwd_pair_frames8_f32: both local convolutions of an interleaved tap tensor [4 B][K][Ho][Wo] in ONE launch (2 B virtual
    samples) against the two calls of savfi_sepconv_bwd_frames8_f32 it replaces -- planar and unit-major taps / gradients, frames of 8-bit
    images, float frames, and a pair of which only ONE frame qualifies (the device then takes the six-product kernel for both)"""
    lib, st = _hip.lib(), _hip.current_stream()
    g = torch.Generator().manual_seed(1000 + Ho)
    frames = []
    for i in range(2):
        if f8 is True or (f8 == "one" and i == 0):
            frames.append(torch.randint(0, 256, (B, 3, Ho + K - 1, Wo + K - 1), generator=g).float().div(255).to(DEV))
        else:
            frames.append(torch.rand(B, 3, Ho + K - 1, Wo + K - 1, generator=g).to(DEV))
    taps = (torch.randn(4 * B, K, Ho, Wo, generator=g) / math.sqrt(K)).to(DEV)
    if u16 & 1:
        taps = _to_unit16(taps)
    gO = torch.randn(B, 3, Ho, Wo, generator=g).to(DEV)
    words = [_words(f) for f in frames]
    plane = K * Ho * Wo * 4
    two = torch.full_like(taps, float('nan'))
    for i, s in ((0, 0), (1, 2)):
        _hip.check(lib.savfi_sepconv_bwd_frames8_f32(frames[i].data_ptr(), taps.data_ptr() + s * plane, taps.data_ptr() + (s + 1) * plane,
                                                     gO.data_ptr(), two.data_ptr() + s * plane, two.data_ptr() + (s + 1) * plane,
                                                     words[i].data_ptr(), B, 3, Ho, Wo, K, 4 * K, u16, st), "two launches")
    one = torch.full_like(taps, float('nan'))
    _hip.check(lib.savfi_sepconv_bwd_pair_frames8_f32(frames[0].data_ptr(), frames[1].data_ptr(), taps.data_ptr(), gO.data_ptr(), one.data_ptr(),
                                                      words[0].data_ptr(), words[1].data_ptr(), B, 3, Ho, Wo, K, u16, st), "pair launch")
    torch.cuda.synchronize()
    assert not torch.isnan(two).any()
    if f8 == "one":
        # frame 0 alone takes the three-product kernel, in the pair both take the six-product kernel: sub-networks 2, 3 are bit-identical,
        # sub-networks 0, 1 agree to fp32 rounding
        t5, o5 = two.view(B, 4, K, Ho, Wo), one.view(B, 4, K, Ho, Wo)
        assert torch.equal(t5[:, 2:], o5[:, 2:])
        assert _rel(o5[:, :2], t5[:, :2]) < 1e-5
    else:
        assert torch.equal(one, two)
    # the forward of the pair: out[b, f] = the local convolution of frame f, bit for bit the two launches'
    out_two = torch.full((2, B, 3, Ho, Wo), float('nan'), device=DEV)
    for i, s in ((0, 0), (1, 2)):
        _hip.check(lib.savfi_sepconv_fwd_frames8_f32(frames[i].data_ptr(), taps.data_ptr() + s * plane, taps.data_ptr() + (s + 1) * plane,
                                                     out_two[i].data_ptr(), words[i].data_ptr(), B, 3, Ho, Wo, K, 4 * K, u16 & 1, st), "two launches")
    out_one = torch.full((B, 2, 3, Ho, Wo), float('nan'), device=DEV)
    _hip.check(lib.savfi_sepconv_fwd_pair_frames8_f32(frames[0].data_ptr(), frames[1].data_ptr(), taps.data_ptr(), out_one.data_ptr(),
                                                      words[0].data_ptr(), words[1].data_ptr(), B, 3, Ho, Wo, K, u16 & 1, st), "pair launch")
    torch.cuda.synchronize()
    assert not torch.isnan(out_two).any()
    if f8 == "one":
        assert torch.equal(out_one[:, 1], out_two[1]) and _rel(out_one[:, 0], out_two[0]) < 1e-5
    else:
        assert torch.equal(out_one.transpose(0, 1), out_two)
    # refusals: a missing pointer, a width that is no multiple of 4
    assert lib.savfi_sepconv_fwd_pair_frames8_f32(frames[0].data_ptr(), frames[1].data_ptr(), None, out_one.data_ptr(),
                                                  words[0].data_ptr(), words[1].data_ptr(), B, 3, Ho, Wo, K, 0, st) == -1
    assert lib.savfi_sepconv_bwd_pair_frames8_f32(frames[0].data_ptr(), None, taps.data_ptr(), gO.data_ptr(), one.data_ptr(),
                                                  words[0].data_ptr(), words[1].data_ptr(), B, 3, Ho, Wo, K, u16, st) == -1
    assert lib.savfi_sepconv_bwd_pair_frames8_f32(frames[0].data_ptr(), frames[1].data_ptr(), taps.data_ptr(), gO.data_ptr(), one.data_ptr(),
                                                  words[0].data_ptr(), words[1].data_ptr(), B, 3, Ho, Wo - 1, K, 0, st) == -3
    assert lib.savfi_sepconv_ws_errors() == 0


def test_a_unit_major_tensor_that_loses_its_tag_is_refused():
    """The taps and their cotangent keep the shape [4N,51,H,W] while their memory is unit-major; producer and consumer agree through a
    tag on the tensor object (hip_ops.tag_layout / require_layout).  A clone, a hook's replacement or a sum of two consumers' gradients
    carries no tag -- or a planar consumer meets a tagged tensor -- and the consumer raises instead of reading scrambled numbers."""
    from meta_interpolation_amd import hip_ops
    B, T, C, Ho, Wo = 1, 4, K, 96, 128
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B * T, C, Ho + 2, Wo + 2, generator=g).to(DEV)
    w = (torch.randn(T, K, C, 3, 3, generator=g) / (3 * math.sqrt(C))).to(DEV)
    b = (torch.randn(T, K, generator=g) * 0.1).to(DEV)
    f0 = torch.randint(0, 256, (B, 3, Ho + K - 1, Wo + K - 1), generator=g).float().div(255).to(DEV)
    gO = torch.randn(B, 3, Ho, Wo, generator=g).to(DEV)
    taps = hip_ops.conv_bias_act_tasks(x, w, b, 1, 0, 1, 1.0, False, None, False, 1)
    assert hip_ops.layout_of(taps) == hip_ops.UNIT16
    with pytest.raises(hip_ops.SavfiLayoutError):
        S.FunctionSepconvPair.apply(f0, f0, taps.clone(), True)            # the clone has the bytes but not the tag
    with pytest.raises(hip_ops.SavfiLayoutError):
        S.FunctionSepconvPair.apply(f0, f0, taps, False)                   # a planar reader of a unit-major tensor
    planar = hip_ops.conv_bias_act_tasks(x, w, b, 1, 0, 1, 1.0, False, None, False, 0)
    with pytest.raises(hip_ops.SavfiLayoutError):
        S.FunctionSepconvPair.apply(f0, f0, planar, True)                  # a unit-major reader of a planar tensor
    # the cotangent: a hook that replaces it (here: by an equal copy) breaks the contract of out_unit16 = 2, and the backward says so
    xs = x.clone().requires_grad_()
    taps = hip_ops.conv_bias_act_tasks(xs, w, b, 1, 0, 1, 1.0, False, None, False, 2)
    taps.register_hook(lambda gr: gr.clone())
    out = S.FunctionSepconvPair.apply(f0, f0, taps, True, True)
    with pytest.raises(hip_ops.SavfiLayoutError):
        out.backward(gO)
    # ... and untouched it goes through
    xs = x.clone().requires_grad_()
    taps = hip_ops.conv_bias_act_tasks(xs, w, b, 1, 0, 1, 1.0, False, None, False, 2)
    S.FunctionSepconvPair.apply(f0, f0, taps, True, True).backward(gO)
    assert xs.grad is not None and torch.isfinite(xs.grad).all()
