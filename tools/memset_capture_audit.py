"""Runs one graphed meta-iteration (training and evaluation graph sets) of every plugin under tools/memset_capture_shim.c and lets
the shim report each hipMemset*Async captured into a hipGraph -- none is acceptable (a memset node only clears in a graph's first
launch on ROCm 7.2: tools/graph_memset_probe.py).

    gcc -O1 -shared -fPIC -o /tmp/memset_capture_shim.so tools/memset_capture_shim.c -ldl
    LD_PRELOAD=/tmp/memset_capture_shim.so python tools/memset_capture_audit.py [model ...] [--size HxW] [--l2f] [--steps S] [--batch B]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('models', nargs='*', default=['cain', 'sepconv', 'voxelflow', 'rrin', 'superslomo'])
ap.add_argument('--size', default='256x448')
ap.add_argument('--l2f', action='store_true')
ap.add_argument('--steps', type=int, default=2)
ap.add_argument('--batch', type=int, default=1)
ap.add_argument('--msl', type=int, default=1)
opt = ap.parse_args()
import torch
from meta_interpolation_amd import synthetic
if opt.l2f:
    from meta_interpolation_amd import graph_inner_loop
    graph_inner_loop.GRAPH_L2F = True
from meta_interpolation_amd.config import default_args
from meta_interpolation_amd.meta_learning_system import MODEL_REGISTRY, SceneAdaptiveInterpolation
H, W = (int(v) for v in opt.size.split('x'))
for model in opt.models:
    print("== %s %dx%d batch %d steps %d l2f %d" % (model, H, W, opt.batch, opt.steps, opt.l2f), file=sys.stderr, flush=True)
    args = default_args(model=model, num_gpu=1, batch_size=opt.batch, number_of_training_steps_per_iter=opt.steps,
                        number_of_evaluation_steps_per_iter=opt.steps, graph_inner_loop=1, attenuate=bool(opt.l2f),
                        use_multi_step_loss_optimization=bool(opt.msl))
    net = MODEL_REGISTRY[model](args, False)
    synthetic.load_seeded_weights(net, model)
    system = SceneAdaptiveInterpolation(args, net=net.cuda())
    frames = [f.cuda() for f in synthetic.septuplet_batch(opt.batch, H, W, model=model)]
    system.run_train_iter(data_batch=frames, epoch=0)
    system.run_validation_iter(data_batch=frames)
    torch.cuda.synchronize()
    print("== %s done (graph sets: %d)" % (model, len(getattr(system, '_graphs', {}))), file=sys.stderr, flush=True)
    del system, net
    torch.cuda.empty_cache()
