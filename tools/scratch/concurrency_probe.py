import os, sys, tempfile
os.environ.setdefault('MIOPEN_USER_DB_PATH', tempfile.mkdtemp(prefix='savfi_cp_'))
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
torch.backends.cudnn.deterministic = True
from meta_interpolation_amd import synthetic
from tests.helpers import build_system
opt, msgd = sys.argv[1], sys.argv[2] == '1'
over = dict(optimizer=opt, metasgd=msgd, inner_lr=1e-4, loss='1*L1', batch_size=4,
            number_of_training_steps_per_iter=2, number_of_evaluation_steps_per_iter=2)
frames = synthetic.septuplet_batch(4, 64, 64, model='cain')
out = []
for streams in (1, 1, 2, 2):
    system = build_system('cain', dict(over, task_streams=streams))
    grads = {}
    system.optimizer.step = lambda *a, **k: grads.update({n: p.grad.detach().clone() for n, p in system.named_parameters() if p.requires_grad and p.grad is not None})
    losses, preds, _ = system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=False)
    torch.cuda.synchronize()
    out.append((losses['loss'].item(), torch.stack([p.squeeze(0) for p in preds]), grads))
for i in range(1, 4):
    a, b = out[0], out[i]
    gd = max(float((a[2][k] - b[2][k]).abs().max() / a[2][k].abs().max().clamp_min(1e-12)) for k in a[2])
    print("run0 vs run%d: loss %.3e  pred max %.3e  per-task pred max %s  grad rel max %.3e" % (
        i, abs(a[0] - b[0]) / abs(a[0]), float((a[1] - b[1]).abs().max()), [round(float((a[1][t] - b[1][t]).abs().max()), 6) for t in range(4)], gd))
