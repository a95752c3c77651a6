"""Feasibility: do two tasks adapted concurrently (2 Python threads, 2 HIP streams) finish sooner than one after the other?"""
import os, sys, tempfile, threading, time
os.environ.setdefault('MIOPEN_USER_DB_PATH', tempfile.mkdtemp(prefix='savfi_ts_'))
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
system._first_order = True


def one_task(t):
    w = system._adapt(frames, t, S, False)
    losses, pred = system._target_pass(frames, t, w, S - 1)
    g = torch.autograd.grad(losses['total'], list(w.values()), allow_unused=True)
    return g


def sequential():
    for t in range(tasks):
        one_task(t)


def threaded(nthreads):
    streams = [torch.cuda.Stream() for _ in range(nthreads)]
    cur = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(cur)

    def run(i):
        with torch.cuda.stream(streams[i]):
            for t in range(i, tasks, nthreads):
                one_task(t)
    th = [threading.Thread(target=run, args=(i,)) for i in range(nthreads)]
    for x in th: x.start()
    for x in th: x.join()
    for s in streams:
        cur.wait_stream(s)


for name, fn in (('sequential', sequential), ('2 threads', lambda: threaded(2)), ('4 threads', lambda: threaded(4)), ('sequential', sequential)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    print("%-12s %.1f ms per meta-batch (4 tasks x 5 steps + target fwd/bwd)" % (name, 1e3 * (time.perf_counter() - t0) / 3), flush=True)
