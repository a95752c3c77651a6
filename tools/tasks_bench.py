"""Per-layer device times of the SepConv network with T tasks in lockstep (support pass: n = 2 samples per task):
savfi task kernels (forward / data gradient / weight gradient) against MIOpen called once per task and as ONE grouped
convolution.  Prints one JSON line per distinct layer and the totals weighted by how often a layer occurs in one pass.

    python tools/tasks_bench.py [--tasks 4] [--n 2] [--iters 10]
"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meta_interpolation_amd import hip_ops  # noqa: E402
from tools.conv_bench import timeit  # noqa: E402

# (Ci, Co, H, W, occurrences per forward, per-task weights?)   -- sepconv/model.py at 256x448 (canvas 384x512)
LAYERS = [
    (6, 32, 384, 512, 1, True), (32, 32, 384, 512, 2, True), (32, 64, 192, 256, 1, True), (64, 64, 192, 256, 2, True),
    (64, 128, 96, 128, 1, True), (128, 128, 96, 128, 2, True), (128, 256, 48, 64, 1, True), (256, 256, 48, 64, 2, True),
    (256, 512, 24, 32, 1, True), (512, 512, 24, 32, 2, True), (512, 512, 12, 16, 3, True),
    (512, 256, 24, 32, 1, True), (256, 256, 24, 32, 2, True), (256, 128, 48, 64, 1, True), (128, 128, 48, 64, 2, True),
    (128, 64, 96, 128, 1, True), (64, 64, 96, 128, 2, True),
    # the plugin's own parameters: shared by all tasks (plain batch of n * T samples)
    (512, 512, 24, 32, 1, False), (256, 256, 48, 64, 1, False), (128, 128, 96, 128, 1, False), (64, 64, 192, 256, 1, False),
    (64, 64, 136, 233, 8, False), (64, 51, 136, 233, 4, False), (51, 51, 258, 450, 4, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tasks", type=int, default=4)
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--miopen", type=int, default=1)
    o = ap.parse_args()
    T, n = o.tasks, o.n
    dev = torch.device("cuda")
    tot = {}
    for (ci, co, h, w, count, per_task) in LAYERS:
        N = n * T
        x = torch.randn(N, ci, h, w, device=dev)
        gy = torch.randn(N, co, h, w, device=dev)
        gflop = 2.0 * 9 * ci * co * h * w * N / 1e9
        row = {"layer": "%d->%d @%dx%d N=%d %s x%d" % (ci, co, h, w, N, "per-task" if per_task else "shared", count), "gflop": round(gflop, 2)}
        if per_task:
            wt = torch.randn(T, co, ci, 3, 3, device=dev) / (3 * ci ** 0.5)
            b = torch.randn(T, co, device=dev)
            row["fwd"] = timeit(lambda: hip_ops.conv3x3_tasks(x, wt, b, 0, 0.0, 1), o.iters)
            row["dgrad"] = timeit(lambda: hip_ops.conv3x3_tasks(gy, wt, None, 1, 1.0, 1), o.iters)
            if ci >= 16:
                row["wgrad"] = timeit(lambda: hip_ops.conv3x3_wgrad_tasks(x, gy, T, 1), o.iters)
            if o.miopen:
                xs = [x[t::T].contiguous() for t in range(T)]
                gs = [gy[t::T].contiguous() for t in range(T)]
                cb = lambda g_, x_, w_, mask, groups=1: torch.ops.aten.convolution_backward(g_, x_, w_, None, [1, 1], [1, 1], [1, 1], False, [0, 0], groups, mask)
                row["mi_fwd_loop"] = timeit(lambda: [F.conv2d(xs[t], wt[t], b[t], padding=1) for t in range(T)], o.iters)
                row["mi_dgrad_loop"] = timeit(lambda: [cb(gs[t], xs[t], wt[t], [True, False, False]) for t in range(T)], o.iters)
                row["mi_wgrad_loop"] = timeit(lambda: [cb(gs[t], xs[t], wt[t], [False, True, False]) for t in range(T)], o.iters)
                xg, gg, wg = x.view(n, T * ci, h, w), gy.view(n, T * co, h, w), wt.view(T * co, ci, 3, 3)
                row["mi_fwd_grouped"] = timeit(lambda: F.conv2d(xg, wg, None, padding=1, groups=T), o.iters)
                row["mi_wgrad_grouped"] = timeit(lambda: cb(gg, xg, wg, [False, True, False], T), max(3, o.iters // 3), reps=3)
        else:
            wt = torch.randn(co, ci, 3, 3, device=dev) / (3 * ci ** 0.5)
            b = torch.randn(co, device=dev)
            row["fwd"] = timeit(lambda: hip_ops.conv3x3(x, wt, b, 0, 0.0, 1), o.iters)
            row["dgrad"] = timeit(lambda: hip_ops.conv3x3(gy, wt, None, 1, 1.0, 1), o.iters)
            row["wgrad"] = timeit(lambda: hip_ops.conv3x3_wgrad(x, gy, 1), o.iters)      # target pass only (outer gradients)
            if o.miopen:
                cb = lambda mask: torch.ops.aten.convolution_backward(gy, x, wt, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, mask)
                row["mi_fwd"] = timeit(lambda: F.conv2d(x, wt, b, padding=1), o.iters)
                row["mi_dgrad"] = timeit(lambda: cb([True, False, False]), o.iters)
                row["mi_wgrad"] = timeit(lambda: cb([False, True, False]), o.iters)
        for k, v in list(row.items()):
            if isinstance(v, float) and k != "gflop":
                row[k] = round(v, 1)
                tot[k + ("" if per_task else "_shared")] = tot.get(k + ("" if per_task else "_shared"), 0.0) + count * v
        row["fwd_TF"] = round(gflop / row["fwd"] * 1e3, 1)
        print(json.dumps(row), flush=True)
    print(json.dumps({"totals_us_per_pass": {k: round(v, 1) for k, v in tot.items()}}), flush=True)


if __name__ == "__main__":
    main()
