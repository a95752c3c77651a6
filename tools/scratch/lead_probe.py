"""How far ahead of the GPU is the host when a support backward returns / when the next forward starts? (no profiler)"""
import os, sys, tempfile, time
os.environ.setdefault('MIOPEN_USER_DB_PATH', tempfile.mkdtemp(prefix='savfi_lp_'))
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

probes = []
def probe(tag):
    e = torch.cuda.Event(enable_timing=True); e.record(); probes.append((tag, time.perf_counter(), e))

orig_update = system.apply_inner_loop_update
def upd(*a, **k):
    probe('backward returned')
    r = orig_update(*a, **k)
    probe('update queued')
    return r
system.apply_inner_loop_update = upd
orig_loss = system._support_loss
def sl(*a, **k):
    probe('support fwd start')
    r = orig_loss(*a, **k)
    probe('support fwd queued')
    return r
system._support_loss = sl

base = torch.cuda.Event(enable_timing=True); base.record(); torch.cuda.synchronize(); t0 = time.perf_counter()
system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=False)
torch.cuda.synchronize()
rows = [(tag, 1e3 * (h - t0), base.elapsed_time(e)) for tag, h, e in probes]
for tag, h, g in rows[:24]:
    print("%-20s host %8.2f ms   gpu reaches it at %8.2f ms   lead %7.2f ms" % (tag, h, g, g - h))
import collections
agg = collections.defaultdict(list)
for tag, h, g in rows: agg[tag].append(g - h)
for tag, v in agg.items(): print(tag, "lead min/median/max ms: %.2f %.2f %.2f" % (min(v), sorted(v)[len(v)//2], max(v)))
