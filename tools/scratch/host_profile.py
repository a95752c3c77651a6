"""Host-side view of one bench iteration (C2): how far ahead of the GPU the host runs, and where its time goes."""
import cProfile, io, os, pstats, sys, tempfile, time
os.environ.setdefault('MIOPEN_USER_DB_PATH', tempfile.mkdtemp(prefix='savfi_hp_'))
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from meta_interpolation_amd import synthetic
from meta_interpolation_amd.config import default_args
from meta_interpolation_amd.meta_learning_system import MODEL_REGISTRY, SceneAdaptiveInterpolation

model, H, W, tasks, S, over = bench.WORKLOADS['c2_sepconv_256x448_b4_s5']
args = default_args(model=model, num_gpu=1, batch_size=tasks, number_of_training_steps_per_iter=S,
                    number_of_evaluation_steps_per_iter=S, fuse_conv_act=1, **over)
dev = torch.device('cuda')
net = MODEL_REGISTRY[model](args, False)
synthetic.load_seeded_weights(net, model)
system = SceneAdaptiveInterpolation(args, net=net.to(dev))
frames = [f.to(dev) for f in synthetic.septuplet_batch(tasks, H, W, model=model)]
for i in range(3):
    system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=False)
torch.cuda.synchronize()
for i in range(3):
    t0 = time.perf_counter()
    system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("iter %d: host enqueue done at %.1f ms, GPU done at %.1f ms" % (i, 1e3 * (t1 - t0), 1e3 * (t2 - t0)), flush=True)
pr = cProfile.Profile()
pr.enable()
for i in range(3):
    system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=False)
torch.cuda.synchronize()
pr.disable()
out = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out')
for key in ('tottime', 'cumulative'):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(60)
    open(os.path.join(out, 'host_profile_%s.txt' % key), 'w').write(s.getvalue())
