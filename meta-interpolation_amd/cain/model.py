"""CAIN plugin (``--model cain``): channel-attention interpolation on pixel-shuffled frames.

Surface and parameter names follow the reference's ``MetaCAIN`` (cain/model.py:10-118):
``encoder.interpolate.{headConv, body.<g>.body.<b>..., tailConv}``; all 494 tensors are fast weights.

    x - mean  ->  reflect-pad to 128  ->  PixelShuffle(1/8) (3 -> 192 ch, savfi HIP kernel)
      ->  MetaInterpolation(5 groups x 12 RCAB, 192 ch)  ->  PixelShuffle(8) (HIP)  ->  crop  ->  + mean
"""
import torch
import torch.nn as nn

from ..model_utils import (InOutPaddings, MetaInterpolation, PixelShuffle, as_view, sub_mean,
                           zero_grad_params)


class Encoder(nn.Module):
    def __init__(self, in_channels=3, depth=3):
        super().__init__()
        self.shuffler = PixelShuffle(1 / 2 ** depth)
        self.interpolate = MetaInterpolation(5, 12, in_channels * (4 ** depth), act=nn.LeakyReLU(0.2, True))

    def forward(self, x1, x2, params=None):
        pv = as_view(params)
        return self.interpolate(self.shuffler(x1), self.shuffler(x2),
                                params=None if pv is None else pv.sub("interpolate"))


class Decoder(nn.Module):
    def __init__(self, depth=3):
        super().__init__()
        self.shuffler = PixelShuffle(2 ** depth)

    def forward(self, feats, params=None):
        return self.shuffler(feats)


class MetaCAIN(nn.Module):
    lockstep_tasks = True     # verified against the sequential loop and the reference fixtures (tests/test_system_gpu.py)
    def __init__(self, depth=3, resume=False):
        super().__init__()
        self.encoder = Encoder(in_channels=3, depth=depth)
        self.decoder = Decoder(depth=depth)
        if resume:
            print('Loading model: pretrained_models/cain_base.pth')
            ckpt = torch.load('pretrained_models/cain_base.pth', map_location='cpu', weights_only=False)      # authors' files pickle an args namespace
            self.load_state_dict({k.replace("module.", ""): v for k, v in ckpt['state_dict'].items()})

    def forward(self, x1, x2, params=None, **kwargs):
        x1, m1 = sub_mean(x1)
        x2, m2 = sub_mean(x2)
        pad_in, pad_out = InOutPaddings(x1)
        pv = as_view(params)
        feats = self.encoder(pad_in(x1), pad_in(x2), params=None if pv is None else pv.sub("encoder"))
        out = pad_out(self.decoder(feats))
        return out + (m1 + m2) / 2

    def zero_grad(self, params=None):
        zero_grad_params(self, params)

    def restore_backup_stats(self):
        pass  # no batch statistics in this model
