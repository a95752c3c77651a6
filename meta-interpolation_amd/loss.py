"""Loss wrapper for the inner/outer objectives: ``'w*TYPE+w*TYPE'`` with TYPE in {L1, MSE}.

Surface follows the reference's ``Loss`` (loss.py:278-350): ``criterion(sr, hr) -> {TYPE: w*loss, ...,
'total': sum}``.  L1 / MSE run on the fused savfi reduction kernel (one launch, no temporaries);
the reference's VGG / GAN / SSIM / SuperSloMo terms are outside this path and are rejected loudly.
"""
import torch.nn as nn

from . import hip_ops

_KERNELS = {'L1': hip_ops.l1_loss, 'MSE': hip_ops.mse_loss}
_KERNELS_PER_SAMPLE = {'L1': hip_ops.l1_loss_per_sample, 'MSE': hip_ops.mse_loss_per_sample}


class Loss(nn.modules.loss._Loss):
    def __init__(self, args):
        super().__init__()
        self.loss = []
        for term in args.loss.split('+'):
            weight, loss_type = term.split('*')
            if loss_type not in _KERNELS:
                raise NotImplementedError(
                    "loss '%s' is outside the inner-loop path built here (only L1, MSE; 'Super' / VGG terms need pretrained VGG16 weights)" % loss_type)
            self.loss.append({'type': loss_type, 'weight': float(weight), 'function': _KERNELS[loss_type]})
        self.cuda_only = True

    def loss_keys(self):
        """Keys of the dict forward() returns (rank-independent: the logging all-reduce is laid out from them)."""
        return [l['type'] for l in self.loss] + ['total']

    def per_sample(self, sr, hr):
        """The same terms for every sample of a batch on its own: {TYPE: [N], 'total': [N]} (tasks adapted in lockstep:
        the reference evaluates the criterion once per task on N=1 tensors, meta_learning_system.py:389-395)."""
        total = 0
        losses = {}
        for l in self.loss:
            eff = l['weight'] * _KERNELS_PER_SAMPLE[l['type']](sr, hr)
            losses[l['type']] = eff
            total = total + eff
        losses['total'] = total
        return losses

    def forward(self, sr, hr, **kwargs):
        total = 0
        losses = {}
        for l in self.loss:
            eff = l['weight'] * l['function'](sr, hr)
            losses[l['type']] = eff
            total = total + eff
        losses['total'] = total
        return losses
