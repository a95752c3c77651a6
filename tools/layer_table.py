"""Per-layer table of the convolution launches of ONE meta-iteration of a bench.py workload (HIP events around every launch).

    python tools/layer_table.py [--workload c2_sepconv_256x448_b4_s5] [--top 40]

One line per (kernel family, layer shape): launches, total ms, average us, direct-equivalent TFLOP/s -- which layers the
iteration's convolution time sits in and how far each is from its ceiling (Winograd on fp32 MFMAs 353.9, split-bf16 direct 416.7).
"""
import argparse
import contextlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from meta_interpolation_amd import _hip, hip_ops, synthetic  # noqa: E402
from meta_interpolation_amd.config import default_args  # noqa: E402
from meta_interpolation_amd.meta_learning_system import MODEL_REGISTRY, SceneAdaptiveInterpolation  # noqa: E402

SUFFIX = [""]


def tag(fn, fmt):
    def wrapped(*a, **k):
        old = SUFFIX[0]
        SUFFIX[0] = fmt(*a, **k)
        try:
            return fn(*a, **k)
        finally:
            SUFFIX[0] = old
    return wrapped


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='c2_sepconv_256x448_b4_s5')
    ap.add_argument('--top', type=int, default=40)
    o = ap.parse_args()
    model, H, W, tasks, S, over = bench.WORKLOADS[o.workload]
    recipe = over.get('weight_recipe')
    over = {k: v for k, v in over.items() if k != 'weight_recipe'}        # the recipe is not a config.py flag (bench.py does the same)
    dev = torch.device('cuda')
    args = default_args(model=model, num_gpu=1, batch_size=tasks, number_of_training_steps_per_iter=S,
                        number_of_evaluation_steps_per_iter=S, **over)
    with contextlib.redirect_stdout(sys.stderr):
        net = MODEL_REGISTRY[model](args, False)
        synthetic.load_seeded_weights(net, model, recipe=recipe)
        system = SceneAdaptiveInterpolation(args, net=net.to(dev))
    frames = [f.to(dev) for f in synthetic.septuplet_batch(tasks, H, W, model=model)]
    sh = lambda x: "x".join(str(int(v)) for v in x.shape)
    hip_ops.convk_tasks_pre = tag(hip_ops.convk_tasks_pre, lambda x, p, T, Ci, Co, K, *a, **k: " %dx%d %d->%d T%d in[%s]" % (K, K, Ci, Co, T, sh(x)))
    hip_ops.conv3x3_tasks_pre = tag(hip_ops.conv3x3_tasks_pre, lambda x, u, T, Ci, Co, *a, **k: " 3x3 %d->%d T%d in[%s]" % (Ci, Co, T, sh(x)))
    hip_ops.convk_wgrad_tasks = tag(hip_ops.convk_wgrad_tasks, lambda x, gz, T, K, *a, **k: " %dx%d %d->%d T%d in[%s]" % (K, K, x.shape[1], gz.shape[1], T, sh(x)))
    hip_ops.conv3x3_wgrad_tasks = tag(hip_ops.conv3x3_wgrad_tasks, lambda x, gz, T, *a, **k: " 3x3 %d->%d T%d in[%s]" % (x.shape[1], gz.shape[1], T, sh(x)))
    hip_ops.conv3x3_wgrad = tag(hip_ops.conv3x3_wgrad, lambda x, gz, *a, **k: " 3x3 %d->%d T1 in[%s]" % (x.shape[1], gz.shape[1], sh(x)))
    orig_launch = _hip.launch
    _hip.launch = lambda name, fn, nbytes=0, flops=0: orig_launch(name + SUFFIX[0], fn, nbytes, flops)
    for _ in range(2):
        system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=False)
    torch.cuda.synchronize()
    timer = _hip.KernelTimer()
    _hip.TIMER = timer
    system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=False)
    _hip.TIMER = None
    summ = timer.summary()
    rows = sorted(summ.items(), key=lambda kv: -kv[1]['total_ms'])
    tot = sum(v['total_ms'] for _, v in rows)
    print("# %s: %.1f ms in timed savfi launches of one meta-iteration" % (o.workload, tot))
    for name, v in rows[:o.top]:
        print("%-62s n=%3d  %7.2f ms  %7.1f us  %s" % (name[:62], v['launches'], v['total_ms'], v['avg_us'],
                                                        ("%.0f TF" % v['direct_TFLOPs']) if v.get('direct_TFLOPs') else
                                                        ("%.0f GB/s" % v['achieved_GBps']) if v.get('achieved_GBps') else ""))


if __name__ == '__main__':
    main()
