import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from meta_interpolation_amd import synthetic, hip_ops, model_utils as mu
from meta_interpolation_amd.sepconv.model import MetaNetwork
for (H, W) in [(260, 272), (256, 256), (256, 448)]:
    fr = synthetic.septuplet_batch(2, H, W, model='sepconv')
    f0, f1 = fr[2].cuda(), fr[4].cuda()
    for windowed in (False, True):
        for fuse in (False, True):
            for wino in (False, True):
                net = MetaNetwork(windowed=windowed); synthetic.load_seeded_weights(net, 'sepconv'); net = net.cuda()
                mu.FUSE_CONV_ACT = fuse; hip_ops.WINOGRAD_CONV = wino
                out = net(f0, f1)
                g = torch.autograd.grad(out.abs().mean(), list(net.parameters()))
                print((H, W), 'windowed', windowed, 'fuse', fuse, 'wino', wino, 'out nan', bool(torch.isnan(out).any()), 'mean', float(out.mean()),
                      'grad nan', any(bool(torch.isnan(x).any()) for x in g), flush=True)
