import os, sys, tempfile, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from meta_interpolation_amd import synthetic, data as D
from meta_interpolation_amd.config import default_args
from meta_interpolation_amd.meta_learning_system import MODEL_REGISTRY, SceneAdaptiveInterpolation
tmp = tempfile.mkdtemp(); os.chdir(tmp)
if os.environ.get("DET"): torch.backends.cudnn.deterministic = True
root = synthetic.write_fake_vimeo(os.path.join(tmp, 'vimeo'))
for variant in ('stager', 'cpu_tensors', 'stager_sync'):
    args = default_args(model='sepconv', num_gpu=1, batch_size=2, number_of_training_steps_per_iter=1, number_of_evaluation_steps_per_iter=1,
                        optimizer='SGD', loss='1*L1', inner_lr=1e-5, dataset='vimeo90k', data_root=root, num_workers=3)
    net = MODEL_REGISTRY['sepconv'](args, False); synthetic.load_seeded_weights(net, 'sepconv')
    system = SceneAdaptiveInterpolation(args, net=net.cuda())
    prov = D.MetaLearningSystemDataLoader(args)
    if variant == 'cpu_tensors':
        prov.stager = None
    n = 0
    for images, meta in prov.get_train_batches(total_batches=4):
        if variant == 'stager_sync':
            torch.cuda.synchronize()
        system.run_train_iter(data_batch=images, epoch=0, do_evaluation=False)      # no host sync inside
        n += 1
    for images, meta in prov.get_val_batches(total_batches=2):
        losses, preds, metrics = system.run_validation_iter(data_batch=images)
    torch.cuda.synchronize()
    print(variant, 'iters', n, 'params finite', all(torch.isfinite(p).all().item() for p in system.parameters()), 'val psnr', metrics['psnr'].avg, flush=True)
