"""RRIN plugin (``--model rrin``): residue refinement interpolation, four U-Nets around a pixel-flow warp.

Surface and parameter names follow the reference's ``MetaRRIN`` / ``MetaUNet`` (rrin/model.py:69-151, rrin/unet.py:101-208):
``{Flow_L, refine_flow, Mask, final}.{down_path.<i>.block.{0,2}, midconv, up_path.<i>.{up.1, conv_block.block.{0,2}}, last}``.

    reflect-pad to 128 -> Flow_L(x0|x1) -> F_0_1, F_1_0 -> F_t_0, F_t_1 (t = 0.5) -> + refine_flow residual
      -> warp x0, x1 (savfi HIP kernel) -> Mask -> sigmoid blend -> + final(x0|x1|blend) -> clamp -> crop

Every convolution is 3x3 / pad 1 and is followed by LeakyReLU(0.1) except the `last` layers: they run through
``MetaConv2dLayer`` (Winograd / MFMA kernels with the activation in the epilogue where a map is large enough).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip_ops
from ..model_utils import InOutPaddings, MetaConv2dLayer, as_view, conv_pair, zero_grad_params

SLOPE = 0.1


def _sub(pv, name):
    return None if pv is None else pv.sub(name)


class MetaUNetConvBlock(nn.Module):
    """(conv3x3 + LeakyReLU(0.1)) x 2; children live at block.0 / block.2 like the reference's MetaSequential."""

    def __init__(self, in_size, out_size):
        super().__init__()
        self.block = nn.ModuleDict({'0': MetaConv2dLayer(in_size, out_size, kernel_size=3, stride=1, padding=1),
                                    '2': MetaConv2dLayer(out_size, out_size, kernel_size=3, stride=1, padding=1)})

    def forward(self, x, params=None):
        pv = _sub(as_view(params), "block")
        return conv_pair(self.block['0'], self.block['2'], x, _sub(pv, "0"), _sub(pv, "2"), SLOPE, SLOPE)


class MetaUNetUpBlock(nn.Module):
    """x2 bilinear (align_corners=False) -> conv3x3 -> cat(centre crop of the bridge) -> conv block."""

    def __init__(self, in_size, out_size):
        super().__init__()
        self.up = nn.ModuleDict({'1': MetaConv2dLayer(in_size, out_size, kernel_size=3, stride=1, padding=1)})
        self.upsample = hip_ops.Upsample2x(align_corners=False)
        self.conv_block = MetaUNetConvBlock(in_size, out_size)

    def forward(self, x, bridge, params=None):
        pv = as_view(params)
        up = self.up['1'](self.upsample(x), params=_sub(_sub(pv, "up"), "1"))
        dy, dx = (bridge.size(2) - up.size(2)) // 2, (bridge.size(3) - up.size(3)) // 2
        crop = bridge[:, :, dy:dy + up.size(2), dx:dx + up.size(3)]
        return self.conv_block(torch.cat((up, crop), 1), params=_sub(pv, "conv_block"))


class MetaUNet(nn.Module):
    def __init__(self, in_channels, n_classes, depth, wf=5):
        super().__init__()
        self.depth = depth
        widths = [2 ** (wf + i) for i in range(depth)]
        self.down_path = nn.ModuleList(MetaUNetConvBlock(c_in, c_out) for c_in, c_out in zip([in_channels] + widths, widths))
        self.midconv = MetaConv2dLayer(widths[-1], widths[-1], kernel_size=3, stride=1, padding=1)
        ups = list(reversed(widths[:-1]))
        self.up_path = nn.ModuleList(MetaUNetUpBlock(c_in, c_out) for c_in, c_out in zip([widths[-1]] + ups, ups))
        self.last = MetaConv2dLayer(widths[0], n_classes, kernel_size=3, stride=1, padding=1)

    def forward(self, x, params=None):
        pv = as_view(params)
        down, up = _sub(pv, "down_path"), _sub(pv, "up_path")
        bridges = []
        for i, block in enumerate(self.down_path):
            x = block(x, params=_sub(down, str(i)))
            if i != self.depth - 1:
                bridges.append(x)
                x = hip_ops.avg_pool2x2(x)
        x = self.midconv(x, params=_sub(pv, "midconv"), act_slope=SLOPE)
        for i, block in enumerate(self.up_path):
            x = block(x, bridges[-i - 1], params=_sub(up, str(i)))
        return self.last(x, params=_sub(pv, "last"))


def warp(img, flow):
    """The reference's warp (rrin/model.py:8-20) on the device kernel; plain ATen on CPU tensors (host-logic tests)."""
    if img.is_cuda:
        return hip_ops.flow_warp(img, flow)
    _, _, H, W = img.shape
    gx = torch.arange(W, dtype=img.dtype).view(1, 1, W) + flow[:, 0]
    gy = torch.arange(H, dtype=img.dtype).view(1, H, 1) + flow[:, 1]
    grid = torch.stack((2 * (gx / W - 0.5), 2 * (gy / H - 0.5)), dim=3)
    return F.grid_sample(img, grid, mode='bilinear', padding_mode='zeros', align_corners=False)


class MetaRRIN(nn.Module):
    lockstep_tasks = True     # verified against the sequential loop and the reference fixtures (tests/test_system_gpu.py)
    def __init__(self, level=3, resume=False):
        super().__init__()
        self.Mask = MetaUNet(16, 2, 4)
        self.Flow_L = MetaUNet(6, 4, 5)
        self.refine_flow = MetaUNet(10, 4, 4)
        self.final = MetaUNet(9, 3, 4)
        if resume:
            print('Loading model: pretrained_models/rrin_base.pth')
            self.load_state_dict(torch.load('pretrained_models/rrin_base.pth', map_location='cpu', weights_only=False))

    def process(self, x0, x1, t, params=None):
        pv = as_view(params)
        x = torch.cat((x0, x1), 1)
        flow = self.Flow_L(x, params=_sub(pv, "Flow_L"))
        f01, f10 = flow[:, :2], flow[:, 2:4]
        ft0 = -(1 - t) * t * f01 + t * t * f10
        ft1 = (1 - t) * (1 - t) * f01 - t * (1 - t) * f10
        residual = self.refine_flow(torch.cat((ft0, ft1, x), 1), params=_sub(pv, "refine_flow"))
        ft0 = ft0 + residual[:, :2]
        ft1 = ft1 + residual[:, 2:4]
        xt0, xt1 = warp(x0, ft0), warp(x1, ft1)
        # the reference calls self.Mask(temp) WITHOUT the fast weights (rrin/model.py:104): the mask net always runs on its
        # own parameters, so its 38 tensors get no gradient after step 0 and leave the fast dict (162 -> 124 live tensors)
        mask = torch.sigmoid(self.Mask(torch.cat((ft0, ft1, x, xt0, xt1), 1), params=None))
        w0, w1 = (1 - t) * mask[:, 0:1], t * mask[:, 1:2]
        return (w0 * xt0 + w1 * xt1) / (w0 + w1 + 1e-8)

    def forward(self, input0, input1, t=0.5, params=None, **kwargs):
        pad_in, pad_out = InOutPaddings(input0)
        input0, input1 = pad_in(input0), pad_in(input1)
        pv = as_view(params)
        blend = self.process(input0, input1, t, params=params)
        final = self.final(torch.cat((input0, input1, blend), 1), params=_sub(pv, "final")) + blend
        return pad_out(final.clamp(0, 1))

    def zero_grad(self, params=None):
        zero_grad_params(self, params)

    def restore_backup_stats(self):
        pass  # no batch statistics in this model
