"""oracle/sepconv_dain.py -- TEST INFRASTRUCTURE (never imported by the product path).

The reference's SECOND statement of the separable local convolution: DAIN's CUDA extension
dain/my_package/SeparableConv/separableconv_cuda_kernel.cu (forward :65-77, backward :113-128).  Restated here loop for loop
(the two filter loops of the kernel are Python loops; the thread grid w_i, h_i, the batch blockIdx.z and the channel loop c_i are
numpy array axes), as the cross-check of oracle/sepconv_ref.c that SURVEY.md 8(c) asks for: the two statements were written
independently by the reference's authors (cupy string kernels vs a compiled extension) and must agree.

    input1 [B, C, H, W]            the padded frame            (H = Ho + K - 1, W = Wo + K - 1)
    input2 [B, K, Ho, Wo]          vertical taps   (indexed by intFilterY, .cu:70)
    input3 [B, K, Ho, Wo]          horizontal taps (indexed by intFilterX, .cu:71)
    output [B, C, Ho, Wo]          out += temp1 * temp2 * temp3                                   (.cu:72)
    backward (.cu:113-128): atomicAdd of gradout*temp2*temp3 into gradinput1[.., h_i + fy, w_i + fx], of gradout*temp1*temp3 into
    gradinput2[.., fy, h_i, w_i] and of gradout*temp1*temp2 into gradinput3[.., fx, h_i, w_i] -- float64 here, so the order of
    the atomics does not matter.
"""
import numpy as np


def forward(input1, input2, input3):
    input1, input2, input3 = (np.asarray(a, dtype=np.float64) for a in (input1, input2, input3))
    B, C, H, W = input1.shape
    K = input2.shape[1]
    ho, wo = H - K + 1, W - K + 1                       # withinYbounds / withinXbounds (.cu:54-55)
    assert input2.shape == (B, K, ho, wo) and input3.shape == (B, K, ho, wo)
    out = np.zeros((B, C, ho, wo))
    for fy in range(K):                                   # intFilterY (.cu:67)
        for fx in range(K):                               # intFilterX (.cu:68)
            temp1 = input1[:, :, fy:fy + ho, fx:fx + wo]                  # (.cu:69)
            temp2 = input2[:, fy][:, None]                                # (.cu:70)
            temp3 = input3[:, fx][:, None]                                # (.cu:71)
            out += temp1 * temp2 * temp3                                  # (.cu:72)
    return out


def backward(input1, input2, input3, gradoutput):
    input1, input2, input3, gradoutput = (np.asarray(a, dtype=np.float64) for a in (input1, input2, input3, gradoutput))
    B, C, H, W = input1.shape
    K = input2.shape[1]
    ho, wo = H - K + 1, W - K + 1
    g1, g2, g3 = np.zeros_like(input1), np.zeros_like(input2), np.zeros_like(input3)
    for fy in range(K):
        for fx in range(K):
            temp1 = input1[:, :, fy:fy + ho, fx:fx + wo]
            temp2 = input2[:, fy][:, None]
            temp3 = input3[:, fx][:, None]
            g1[:, :, fy:fy + ho, fx:fx + wo] += gradoutput * temp2 * temp3           # (.cu:120-121)
            g2[:, fy] += (gradoutput * temp1 * temp3).sum(1)                         # (.cu:122-123), summed over c_i
            g3[:, fx] += (gradoutput * temp1 * temp2).sum(1)                         # (.cu:124-125)
    return g1, g2, g3
