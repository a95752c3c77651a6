"""Deep Voxel Flow plugin (``--model voxelflow``).

Surface and parameter names follow the reference's ``MetaVoxelFlow``
(voxelflow/core/models/voxel_flow.py:231-534): conv1/2/3, bottleneck, deconv1/2/3 (bias-free,
5x5 / 3x3), ``*_bn`` BatchNorm2d always in eval mode, conv4 (64 -> 3, with bias), tanh, then the
warp-and-blend.  BN affine tensors sit in the inner-loop dict (their names lack 'norm_layer') but the
BN layers always use their own parameters, exactly as the reference (:379,385,...).

The reference's warp tail (:471-509: meshgrid on the CPU + H2D copy every call, 2 grid_samples and
~15 elementwise launches) is ONE savfi HIP kernel forward and one backward (hip_ops.voxel_warp_blend).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ....model_utils import MetaConv2dLayer, as_view, zero_grad_params
from .... import hip_ops

BatchNorm2d = nn.BatchNorm2d

# name, in, out, kernel  (encoder -> bottleneck -> decoder); every conv is followed by <name>_bn
_TRUNK = [("conv1", 6, 64, 5), ("conv2", 64, 128, 5), ("conv3", 128, 256, 3), ("bottleneck", 256, 256, 3),
          ("deconv1", 512, 256, 3), ("deconv2", 384, 128, 5), ("deconv3", 192, 64, 5)]


class MetaVoxelFlow(nn.Module):
    # Round 3: every layer (five 5x5, three 3x3) runs on the direct split-bf16 convolution kernels in `precise` mode with per-task
    # filter sets (hip_ops.convk_*), so the tasks of a meta-batch advance in lockstep like the other plugins (config C3:
    # 208 steps/s from graphs on four task streams, 254 with its 8 tasks in one lockstep group).  BatchNorm is frozen in eval
    # mode (no coupling between samples), fast weights are read through MetaConv2dLayer only.
    lockstep_tasks = True

    def __init__(self, config, resume=False):
        super().__init__()
        self.config = config
        self.input_mean = [0.5 * 255] * 3
        self.input_std = [0.5 * 255] * 3
        self.syn_type = 'inter'
        self.relu = nn.ReLU(inplace=True)
        self.pool = nn.MaxPool2d(kernel_size=2, stride=2)
        for name, cin, cout, k in _TRUNK:
            setattr(self, name, MetaConv2dLayer(cin, cout, kernel_size=k, stride=1, padding=k // 2, use_bias=False, direct=True))
            setattr(self, name + "_bn", BatchNorm2d(cout, momentum=0.9997))
        self.conv4 = MetaConv2dLayer(64, 3, kernel_size=5, stride=1, padding=2, direct=True)

        for m in self.modules():
            if isinstance(m, MetaConv2dLayer):
                m.weight.data.normal_(0, 0.01)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        if resume:
            print('Loading model: pretrained_models/voxelflow_ft.pth')
            self.load_state_dict(torch.load('pretrained_models/voxelflow_ft.pth', map_location='cpu', weights_only=False)['state_dict'])
        self.fix_batchnorm_parameters()
        for attr, val in (('mult_conv_w', [1, 1]), ('mult_conv_b', [2, 0]), ('mult_bn', [1, 1])):
            setattr(self.config, attr, val)

    # BN statistics are frozen in every mode
    def train(self, mode=True):
        super().train(mode)
        self.fix_batchnorm_parameters()
        return self

    def fix_batchnorm_parameters(self):
        for m in self.modules():
            if isinstance(m, BatchNorm2d):
                m.eval()

    def get_optim_policies(self):
        """Three outer-optimizer parameter groups: conv weights, conv biases, BN affine (reference :307-350)."""
        weight, bias, bn = [], [], []
        for m in self.modules():
            if isinstance(m, MetaConv2dLayer):
                weight.append(m.weight)
                if m.bias is not None:
                    bias.append(m.bias)
            elif isinstance(m, BatchNorm2d):
                bn.extend(m.parameters())
        cfg = self.config
        return [
            {'params': weight, 'lr_mult': cfg.mult_conv_w[0], 'decay_mult': cfg.mult_conv_w[1], 'name': 'model weight'},
            {'params': bias, 'lr_mult': cfg.mult_conv_b[0], 'decay_mult': cfg.mult_conv_b[1], 'name': 'model bias'},
            {'params': bn, 'lr_mult': cfg.mult_bn[0], 'decay_mult': cfg.mult_bn[1], 'name': 'model bn scale/shift'},
        ]

    def _block(self, name, x, pv):
        x = getattr(self, name)(x, params=None if pv is None else pv.sub(name))
        return self.relu(getattr(self, name + "_bn")(x))

    def forward(self, x0, x1, syn_type='inter', params=None, **kwargs):
        x = torch.cat([x0, x1], dim=1)
        h, w = x.size(2), x.size(3)
        rnd = lambda n: 0 if n == ((n >> 6) << 6) else (((n >> 6) + 1) << 6) - n
        pw, ph = rnd(w), rnd(h)
        left, top = pw // 2, ph // 2
        frames = F.pad(x, (left, pw - left, top, ph - top), mode='reflect') if (pw or ph) else x

        pv = as_view(params)
        c1 = self._block("conv1", frames, pv)
        c2 = self._block("conv2", self.pool(c1), pv)
        c3 = self._block("conv3", self.pool(c2), pv)
        y = self._block("bottleneck", self.pool(c3), pv)
        up = lambda t: hip_ops.upsample_bilinear2x(t, align_corners=False)
        y = self._block("deconv1", torch.cat([up(y), c3], dim=1), pv)
        y = self._block("deconv2", torch.cat([up(y), c2], dim=1), pv)
        y = self._block("deconv3", torch.cat([up(y), c1], dim=1), pv)
        x3 = torch.tanh(self.conv4(y, params=None if pv is None else pv.sub("conv4")))

        out = hip_ops.voxel_warp_blend(frames, x3)
        return out[:, :, top:top + h, left:left + w]

    def zero_grad(self, params=None):
        zero_grad_params(self, params)

    def restore_backup_stats(self):
        pass  # BN statistics are frozen
