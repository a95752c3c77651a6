"""Evaluation helpers on the adaptation path: AverageMeter, quantize / PSNR / SSIM (the parity
metric), checkpoint save / load with the reference's key layout.

Reference: utils.py:34-118 (checkpoints), :135-150 (AverageMeter), :171-204 (metrics);
pytorch_msssim/__init__.py:19-75 (ssim).  Image / video writers and tensorboard are out of scope.
"""
import math
import os
import shutil

import torch
import torch.nn.functional as F


class AverageMeter(object):
    """Running value / sum / count / average."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def quantize(img, rgb_range=255):
    return img.mul(255 / rgb_range).clamp(0, 255).round()


def calc_psnr(pred, gt, mask=None):
    """PSNR of 0..255-quantised tensors: -10 log10(mean(((p-g)/255)^2) + 1e-8)  (reference :177-186)."""
    diff = (pred - gt).div(255)
    if mask is not None:
        mse = diff.pow(2).sum() / (3 * mask.sum())
    else:
        mse = diff.pow(2).mean() + 1e-8
    return -10 * math.log10(mse)


def _gauss_window(size, channel, device, sigma=1.5):
    # taps evaluated in double precision, then rounded (as the reference's math.exp list does)
    g = torch.tensor([math.exp(-((i - size // 2) ** 2) / (2.0 * sigma ** 2)) for i in range(size)],
                     dtype=torch.float32)
    g = (g / g.sum()).unsqueeze(1)
    w2 = g.mm(g.t()).unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, size, size).contiguous().to(device)


def ssim(img1, img2, window_size=11, val_range=255):
    """Gaussian-window SSIM, valid convolution, C1/C2 from val_range (pytorch_msssim/__init__.py:19-75)."""
    _, channel, height, width = img1.size()
    win = _gauss_window(min(window_size, height, width), channel, img1.device)
    conv = lambda t: F.conv2d(t, win, padding=0, groups=channel)
    mu1, mu2 = conv(img1), conv(img2)
    mu1_sq, mu2_sq, mu12 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = conv(img1 * img1) - mu1_sq
    s2 = conv(img2 * img2) - mu2_sq
    s12 = conv(img1 * img2) - mu12
    C1, C2 = (0.01 * val_range) ** 2, (0.03 * val_range) ** 2
    v1, v2 = 2.0 * s12 + C2, s1 + s2 + C2
    return (((2 * mu12 + C1) * v1) / ((mu1_sq + mu2_sq + C1) * v2)).mean()


def calc_metrics(im_pred, im_gt, mask=None):
    """(PSNR float, SSIM 0-dim tensor) of two [3,H,W] images in [0,1]  (reference :189-204)."""
    q_pred = quantize(im_pred.data, rgb_range=1.)
    q_gt = quantize(im_gt.data, rgb_range=1.)
    if mask is not None:
        q_pred, q_gt = q_pred * mask, q_gt * mask
    psnr = calc_psnr(q_pred, q_gt, mask=mask)
    return psnr, ssim(q_pred.unsqueeze(0), q_gt.unsqueeze(0), val_range=255)


# ---------------------------------------------------------------------------------------------
# checkpoints: {'epoch', 'arch', 'state_dict', 'best_PSNR'} under checkpoint/<exp_name>/
# ---------------------------------------------------------------------------------------------
def save_checkpoint(state, is_best, exp_name, filename='checkpoint.pth'):
    directory = os.path.join('checkpoint', exp_name)
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, filename)
    torch.save(state, path)
    if is_best:
        shutil.copyfile(path, os.path.join(directory, 'model_best.pth'))


def lossy_load_state_dict(model, state_dict, verbose=False):
    """Copy every entry whose name and shape match; report the rest (reference :89-107)."""
    own = model.state_dict()
    loaded, skipped = [], []
    for name, value in state_dict.items():
        name = name[len('module.'):] if name.startswith('module.') else name
        if name in own and own[name].shape == value.shape:
            own[name].copy_(value)
            loaded.append(name)
        else:
            skipped.append(name)
    if verbose and skipped:
        print('lossy_load_state_dict: skipped', skipped)
    return loaded, skipped


def update_lr(optimizer, lr):
    for group in optimizer.param_groups:
        group['lr'] = lr


def load_checkpoint(args, model, optimizer=None, fix_loaded=False):
    """Resume from checkpoint/<resume_exp or exp_name>/{checkpoint,model_best}.pth (reference :34-86): keys are
    filtered by name and shape; args.start_epoch comes from the file (0 when resuming ANOTHER experiment's weights);
    the optimizer state is restored only if every model tensor was found, outside test mode."""
    if getattr(args, 'resume_exp', None) is None:
        args.resume_exp = args.exp_name
    name = 'model_best.pth' if getattr(args, 'mode', 'train') in ('val', 'test') else 'checkpoint.pth'
    path = os.path.join('checkpoint', args.resume_exp, name)
    print("loading checkpoint %s" % path)
    ckpt = torch.load(path, map_location='cpu', weights_only=False)
    args.start_epoch = ckpt.get('epoch', 0)
    if args.resume_exp != args.exp_name:
        args.start_epoch = 0
    with torch.no_grad():
        loaded, skipped = lossy_load_state_dict(model, ckpt['state_dict'], verbose=True)
    mismatch = bool(skipped) or len(model.state_dict()) > len(loaded)
    if not mismatch and optimizer is not None and getattr(args, 'mode', 'train') != 'test' and 'optimizer' in ckpt:
        optimizer.load_state_dict(ckpt['optimizer'])
        if hasattr(args, 'lr'):
            update_lr(optimizer, args.lr)
    if fix_loaded:
        params = dict(model.named_parameters())
        for name in loaded:
            if name in params:
                params[name].requires_grad = False
    print("loaded checkpoint %s" % path)
    return ckpt


# ---------------------------------------------------------------------------------------------
# image writer (reference :276-285): [C,H,W] or [H,W] tensor in [0,1] -> 8-bit PNG
# ---------------------------------------------------------------------------------------------
def save_image(img, path):
    from PIL import Image
    q = quantize(img.detach().mul(255)).cpu().numpy().astype('uint8')
    if img.dim() == 2:
        im = Image.fromarray(q, 'L')
    elif img.dim() == 3:
        im = Image.fromarray(q.transpose(1, 2, 0), 'RGB')
    else:
        return
    im.save(path)
