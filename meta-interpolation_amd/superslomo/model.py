"""Super SloMo plugin (``--model superslomo``): two U-Nets (flow computation, arbitrary-time flow interpolation)
around the pixel-flow backward warp.

Surface and parameter names follow the reference's ``MetaSuperSloMo`` / ``MetaUNet`` (superslomo/model.py:457-670):
``{flowComp, arbTimeFlowIntrp}.{conv1, conv2, down<1..5>.{conv1,conv2}, up<1..5>.{conv1,conv2}, conv3}``.
``forward`` returns ``(frame, extras)`` with the flows and warped frames the reference's 'Super' loss consumes.

    reflect-pad to 64 -> flowComp(I0|I1) -> F_0_1, F_1_0 -> F_t_0, F_t_1 -> warp I0, I1 (savfi HIP kernel)
      -> arbTimeFlowIntrp(20 ch) -> flow residuals + visibility -> warp again -> visibility-weighted blend -> crop
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip_ops
from ..model_utils import MetaConv2dLayer, _pad_to_multiple, as_view, conv_pair, zero_grad_params
from ..rrin.model import warp

SLOPE = 0.1
# the 7 intermediate time stamps between I0 and I1 (reference :308); `ind` selects one (3 -> t = 0.5)
T_GRID = np.linspace(0.125, 0.875, 7)


def _sub(pv, name):
    return None if pv is None else pv.sub(name)


def _conv(c_in, c_out, k):
    return MetaConv2dLayer(in_channels=c_in, out_channels=c_out, kernel_size=k, stride=1, padding=(k - 1) // 2)


class down(nn.Module):
    """avg-pool 2 -> (conv k x k + LeakyReLU(0.1)) x 2"""

    def __init__(self, inChannels, outChannels, filterSize):
        super().__init__()
        self.conv1 = _conv(inChannels, outChannels, filterSize)
        self.conv2 = _conv(outChannels, outChannels, filterSize)

    def forward(self, x, params=None):
        pv = as_view(params)
        return conv_pair(self.conv1, self.conv2, hip_ops.avg_pool2x2(x), _sub(pv, "conv1"), _sub(pv, "conv2"), SLOPE, SLOPE)


class up(nn.Module):
    """x2 bilinear (align_corners=False) -> conv3x3 + LeakyReLU -> cat(skip) -> conv3x3 + LeakyReLU"""

    def __init__(self, inChannels, outChannels):
        super().__init__()
        self.conv1 = _conv(inChannels, outChannels, 3)
        self.conv2 = _conv(2 * outChannels, outChannels, 3)
        self.upsample = hip_ops.Upsample2x(align_corners=False)

    def forward(self, x, skpCn, params=None):
        pv = as_view(params)
        x = self.conv1(self.upsample(x), params=_sub(pv, "conv1"), act_slope=SLOPE)
        return self.conv2(torch.cat((x, skpCn), 1), params=_sub(pv, "conv2"), act_slope=SLOPE)


class MetaUNet(nn.Module):
    def __init__(self, inChannels, outChannels):
        super().__init__()
        self.conv1 = _conv(inChannels, 32, 7)
        self.conv2 = _conv(32, 32, 7)
        self.down1 = down(32, 64, 5)
        self.down2 = down(64, 128, 3)
        self.down3 = down(128, 256, 3)
        self.down4 = down(256, 512, 3)
        self.down5 = down(512, 512, 3)
        self.up1 = up(512, 512)
        self.up2 = up(512, 256)
        self.up3 = up(256, 128)
        self.up4 = up(128, 64)
        self.up5 = up(64, 32)
        self.conv3 = _conv(32, outChannels, 3)

    def forward(self, x, params=None):
        pv = as_view(params)
        skips = [conv_pair(self.conv1, self.conv2, x, _sub(pv, "conv1"), _sub(pv, "conv2"), SLOPE, SLOPE)]
        for name in ("down1", "down2", "down3", "down4"):
            skips.append(getattr(self, name)(skips[-1], params=_sub(pv, name)))
        x = self.down5(skips[-1], params=_sub(pv, "down5"))
        for name in ("up1", "up2", "up3", "up4", "up5"):
            x = getattr(self, name)(x, skips.pop(), params=_sub(pv, name))
        return self.conv3(x, params=_sub(pv, "conv3"), act_slope=SLOPE)


class backWarp(nn.Module):
    """I0 <- backwarp(I1, F_0_1): bilinear sampling at the flow target (reference :231-307)."""

    def __init__(self, W, H, device=None):
        super().__init__()
        self.W, self.H = W, H

    def forward(self, img, flow):
        return warp(img, flow)


def getFlowCoeff(indices, device):
    """C00, C01, C10, C11 of  F_t_0 = C00 F_0_1 + C01 F_1_0,  F_t_1 = C10 F_0_1 + C11 F_1_0  as [B,1,1,1] tensors."""
    t = T_GRID[np.asarray(indices)]
    coeffs = (-(1 - t) * t, t * t, (1 - t) * (1 - t), -(1 - t) * t)
    return tuple(torch.tensor(c, dtype=torch.float32, device=device).view(-1, 1, 1, 1) for c in coeffs)


def getWarpCoeff(indices, device):
    """C0 = 1 - t, C1 = t of the visibility-weighted blend."""
    t = T_GRID[np.asarray(indices)]
    return tuple(torch.tensor(c, dtype=torch.float32, device=device).view(-1, 1, 1, 1) for c in (1 - t, t))


def _time_coefficients(ind):
    """(C00, C01, C10, C11), (C0, C1) as Python floats rounded to fp32, for one time stamp shared by the whole batch (the
    plugin is always called with a scalar `ind`): no host-to-device copy per forward, so the pass can be captured in a hipGraph."""
    t = T_GRID[int(ind)]
    f32 = lambda v: float(np.float32(v))
    return (f32(-(1 - t) * t), f32(t * t), f32((1 - t) * (1 - t)), f32(-(1 - t) * t)), (f32(1 - t), f32(t))


class MetaSuperSloMo(nn.Module):
    lockstep_tasks = True     # verified against the sequential loop and the reference fixtures (tests/test_system_gpu.py)
    def __init__(self, device=None, resume=False):
        super().__init__()
        self.device = device
        self.flowComp = MetaUNet(6, 4)
        self.arbTimeFlowIntrp = MetaUNet(20, 5)
        if resume:
            print('Loading model: pretrained_models/superslomo_base.pth')
            ckpt = torch.load('pretrained_models/superslomo_base.pth', map_location='cpu', weights_only=False)
            self.flowComp.load_state_dict(ckpt['state_dictFC'])
            self.arbTimeFlowIntrp.load_state_dict(ckpt['state_dictAT'])

    def forward(self, I0, I1, ind=3, params=None, **kwargs):
        (c00, c01, c10, c11), (c0, c1) = _time_coefficients(ind)
        pw, ph = _pad_to_multiple(I0.size(3), 6), _pad_to_multiple(I0.size(2), 6)
        left, top = pw // 2, ph // 2
        pad_in = nn.ReflectionPad2d([left, pw - left, top, ph - top])
        crop = lambda x: x[:, :, top:x.size(2) - (ph - top), left:x.size(3) - (pw - left)]
        I0, I1 = pad_in(I0), pad_in(I1)
        pv = as_view(params)

        flows = self.flowComp(torch.cat((I0, I1), dim=1), params=_sub(pv, "flowComp"))
        F_0_1, F_1_0 = flows[:, :2], flows[:, 2:]
        F_t_0 = c00 * F_0_1 + c01 * F_1_0
        F_t_1 = c10 * F_0_1 + c11 * F_1_0
        g_I0_F_t_0, g_I1_F_t_1 = warp(I0, F_t_0), warp(I1, F_t_1)
        intrp = self.arbTimeFlowIntrp(torch.cat((I0, I1, F_0_1, F_1_0, F_t_1, F_t_0, g_I1_F_t_1, g_I0_F_t_0), dim=1),
                                      params=_sub(pv, "arbTimeFlowIntrp"))
        F_t_0_f = intrp[:, :2] + F_t_0
        F_t_1_f = intrp[:, 2:4] + F_t_1
        V_t_0 = torch.sigmoid(intrp[:, 4:5])
        V_t_1 = 1 - V_t_0
        g0, g1 = warp(I0, F_t_0_f), warp(I1, F_t_1_f)
        Ft_p = (c0 * V_t_0 * g0 + c1 * V_t_1 * g1) / (c0 * V_t_0 + c1 * V_t_1)
        warped_I0, warped_I1 = warp(I0, F_1_0), warp(I1, F_0_1)
        return crop(Ft_p), {'bidirectional_flow': (crop(F_0_1), crop(F_1_0)),
                            'warped_intermediate_frames': (crop(g_I0_F_t_0), crop(g_I1_F_t_1)),
                            'warped_input_frames': (crop(warped_I0), crop(warped_I1))}

    def zero_grad(self, params=None):
        zero_grad_params(self, params)

    def restore_backup_stats(self):
        pass  # no batch statistics in this model
