import os, sys, tempfile, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from meta_interpolation_amd import synthetic
from meta_interpolation_amd.config import default_args
from meta_interpolation_amd.data import MetaLearningSystemDataLoader
from meta_interpolation_amd.meta_learning_system import MODEL_REGISTRY, SceneAdaptiveInterpolation
tmp = tempfile.mkdtemp(); os.chdir(tmp)
if os.environ.get("DET"): torch.backends.cudnn.deterministic = True
if os.environ.get("POISON"):
    junk = [torch.full((1 << 26,), float("nan"), device="cuda") for _ in range(8)]
    del junk
root = synthetic.write_fake_vimeo(os.path.join(tmp, 'vimeo'))
args = default_args(model='sepconv', num_gpu=1, batch_size=2, number_of_training_steps_per_iter=1, number_of_evaluation_steps_per_iter=1,
                    optimizer='SGD', loss='1*L1', inner_lr=1e-5, dataset='vimeo90k', data_root=root, num_workers=3)
net = MODEL_REGISTRY['sepconv'](args, False); synthetic.load_seeded_weights(net, 'sepconv')
system = SceneAdaptiveInterpolation(args, net=net.cuda())
prov = MetaLearningSystemDataLoader(args)
fin = lambda: all(torch.isfinite(p).all().item() for p in system.parameters())
for i, (images, meta) in enumerate(prov.get_train_batches(total_batches=3)):
    print('batch', i, [tuple(t.shape) for t in images[:1]], 'min/max', float(images[0].min()), float(images[0].max()), 'nan in', any(bool(torch.isnan(t).any()) for t in images))
    losses, preds, metrics = system.run_train_iter(data_batch=images, epoch=0, do_evaluation=True)
    gn = max([float(p.grad.abs().max()) for p in system.parameters() if p.grad is not None] + [-1.0])
    print('  loss', float(losses['loss']), 'psnr', metrics['psnr'].avg, 'params finite', fin(), 'max |grad|', gn)
for i, (images, meta) in enumerate(prov.get_val_batches(total_batches=2)):
    losses, preds, metrics = system.run_validation_iter(data_batch=images)
    print('val', i, tuple(images[0].shape), 'loss', float(losses['loss']), 'psnr', metrics['psnr'].avg, 'pred nan', bool(torch.isnan(preds[0]).any()))

print('--- through ExperimentBuilder')
from meta_interpolation_amd.experiment_builder import ExperimentBuilder
args2 = default_args(model='sepconv', num_gpu=1, batch_size=2, number_of_training_steps_per_iter=1, number_of_evaluation_steps_per_iter=1,
                     optimizer='SGD', loss='1*L1', inner_lr=1e-5, dataset='vimeo90k', data_root=root, total_iter_per_epoch=2, max_epoch=1,
                     exp_name='vimeo_e2e', num_workers=3)
net2 = MODEL_REGISTRY['sepconv'](args2, False); synthetic.load_seeded_weights(net2, 'sepconv')
system2 = SceneAdaptiveInterpolation(args2, net=net2.cuda())
fin2 = lambda: all(torch.isfinite(p).all().item() for p in system2.parameters())
orig_t, orig_v = system2.run_train_iter, system2.run_validation_iter
def wt(**kw):
    imgs = kw['data_batch']
    r = orig_t(**kw); print('  train it: in nan', any(bool(torch.isnan(t).any()) for t in imgs), 'in max', max(float(t.abs().max()) for t in imgs), 'loss', float(r[0]['loss']), 'finite', fin2()); return r
def wv(**kw):
    imgs = kw['data_batch']
    r = orig_v(**kw); print('  val it: in nan', any(bool(torch.isnan(t).any()) for t in imgs), 'in max', max(float(t.abs().max()) for t in imgs), 'loss', float(r[0]['loss']), 'psnr', r[2]['psnr'].avg); return r
system2.run_train_iter, system2.run_validation_iter = wt, wv
eb = ExperimentBuilder(args2, MetaLearningSystemDataLoader, system2)
eb.run_experiment()
