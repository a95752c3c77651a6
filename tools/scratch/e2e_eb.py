import os, sys, tempfile, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from meta_interpolation_amd import synthetic, data as D
from meta_interpolation_amd.config import default_args
from meta_interpolation_amd.experiment_builder import ExperimentBuilder
from meta_interpolation_amd.meta_learning_system import MODEL_REGISTRY, SceneAdaptiveInterpolation
if os.environ.get("DET"): torch.backends.cudnn.deterministic = True
tmp = tempfile.mkdtemp(); os.chdir(tmp)
root = synthetic.write_fake_vimeo(os.path.join(tmp, 'vimeo'))
for variant in sys.argv[1:]:
    args = default_args(model='sepconv', num_gpu=1, batch_size=2, number_of_training_steps_per_iter=1, number_of_evaluation_steps_per_iter=1,
                        optimizer='SGD', loss='1*L1', inner_lr=1e-5, dataset='vimeo90k', data_root=root, total_iter_per_epoch=2, max_epoch=1,
                        exp_name='e2e_' + variant, num_workers=3)
    net = MODEL_REGISTRY['sepconv'](args, False); synthetic.load_seeded_weights(net, 'sepconv')
    system = SceneAdaptiveInterpolation(args, net=net.cuda())
    eb = ExperimentBuilder(args, D.MetaLearningSystemDataLoader, system)
    if variant == 'nostager':
        eb.data.stager = None
    if variant == 'stager_clone':
        orig = eb.data._batches
        eb.data._batches = lambda mode: (([t.clone() for t in im], me) for im, me in orig(mode))
    if variant == 'stager_sync':
        orig = eb.data._batches
        def synced(mode):
            for im, me in orig(mode):
                torch.cuda.synchronize(); yield im, me
        eb.data._batches = synced
    if os.environ.get('POISON'):
        junk = [torch.full((1 << 26,), float('nan'), device='cuda') for _ in range(12)]
        del junk
    eb.run_experiment()
    torch.cuda.synchronize()
    print(variant, 'params finite', all(torch.isfinite(p).all().item() for p in system.parameters()), flush=True)
