"""Direct K x K convolution on split-bf16 MFMAs (savfi_convk_*): correctness against a float64 CPU convolution and device
times against the Winograd kernel (3x3) and MIOpen, on the layer shapes of the plugins.

    python tools/convk_bench.py [--check] [--time] [--iters 10]
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

# (K, Ci, Co, H, W, pad, T, N)
CHECK = [
    (3, 6, 32, 20, 40, 1, 1, 1), (3, 32, 32, 24, 64, 1, 2, 4), (3, 64, 51, 17, 33, 1, 1, 2), (3, 51, 51, 18, 50, 0, 1, 1),
    (3, 128, 64, 12, 16, 1, 4, 8), (3, 24, 40, 9, 70, 1, 1, 1), (3, 192, 192, 16, 16, 1, 1, 2),
    (5, 6, 64, 32, 32, 2, 1, 2), (5, 64, 128, 16, 48, 2, 2, 2), (5, 192, 64, 12, 20, 2, 1, 1), (5, 64, 3, 24, 40, 2, 1, 2),
    (7, 6, 32, 24, 40, 3, 1, 1), (7, 32, 32, 16, 36, 3, 2, 2), (5, 32, 64, 20, 20, 2, 1, 1), (7, 20, 2, 10, 12, 3, 1, 1),
]
# (K, Ci, Co, H, W, T, N, label)
TIME = [
    (3, 32, 32, 384, 512, 4, 8, "sepconv"), (3, 64, 64, 192, 256, 4, 8, "sepconv"), (3, 128, 128, 96, 128, 4, 8, "sepconv"),
    (3, 256, 256, 48, 64, 4, 8, "sepconv"), (3, 512, 512, 24, 32, 4, 8, "sepconv"), (3, 512, 512, 12, 16, 4, 8, "sepconv"),
    (3, 64, 64, 137, 233, 1, 8, "sepconv subnet"), (3, 51, 51, 258, 450, 1, 8, "sepconv subnet"), (3, 6, 32, 384, 512, 4, 8, "sepconv"),
    (3, 192, 192, 96, 160, 1, 2, "cain 720p"), (3, 192, 192, 96, 160, 1, 1, "cain 720p"), (3, 192, 192, 16, 16, 1, 2, "cain 64x64"),
    (5, 6, 64, 256, 256, 1, 2, "voxelflow"), (5, 64, 128, 128, 128, 1, 2, "voxelflow"), (5, 384, 128, 128, 128, 1, 2, "voxelflow"),
    (5, 192, 64, 256, 256, 1, 2, "voxelflow"), (5, 64, 3, 256, 256, 1, 2, "voxelflow"), (3, 128, 256, 64, 64, 1, 2, "voxelflow"),
    (7, 6, 32, 256, 448, 1, 2, "superslomo"), (7, 32, 32, 256, 448, 1, 2, "superslomo"), (5, 32, 64, 128, 224, 1, 2, "superslomo"),
]


def reference(x, w, b, pad, T, slope, dtype=torch.float64):
    x, w = x.cpu().to(dtype), w.cpu().to(dtype)
    outs = []
    for n in range(x.shape[0]):
        t = n % T
        outs.append(F.conv2d(x[n:n + 1], w[t], None if b is None else b[t].cpu().to(dtype), padding=pad))
    z = torch.cat(outs, 0)
    return torch.where(z > 0, z, z * slope)


def check():
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    worst = 0.0
    for (K, ci, co, H, W, pad, T, N) in CHECK:
        x = torch.randn(N, ci, H, W, generator=g).to(dev)
        w = (torch.randn(T, co, ci, K, K, generator=g) / (K * ci ** 0.5)).to(dev)
        b = torch.randn(T, co, generator=g).to(dev)
        pf, pb = hip_ops.convk_filters(w, True, True)
        y = hip_ops.convk_tasks_pre(x, pf, T, ci, co, K, b, 0, 0.2, pad)
        ref = reference(x, w, b, pad, T, 0.2)
        ref32 = reference(x, w, b, pad, T, 0.2, torch.float32).double()
        err = ((y.cpu().double() - ref).abs().max() / ref.abs().max()).item()
        err32 = ((ref32 - ref).abs().max() / ref.abs().max()).item()
        # data gradient: gx = conv_transpose(gy, w)
        Ho, Wo = H + 2 * pad - K + 1, W + 2 * pad - K + 1
        gy = torch.randn(N, co, Ho, Wo, generator=g).to(dev)
        gx = hip_ops.convk_tasks_pre(gy, pb, T, ci, co, K, None, 1, 1.0, pad)
        gref = torch.cat([F.conv_transpose2d(gy[n:n + 1].cpu().double(), w[n % T].cpu().double(), padding=pad) for n in range(N)], 0)
        gerr = ((gx.cpu().double() - gref).abs().max() / gref.abs().max()).item()
        # weight gradient
        gw = hip_ops.convk_wgrad_tasks(x, gy, T, K, pad)
        wref = torch.stack([torch.nn.grad.conv2d_weight(x[t::T].cpu().double(), (co, ci, K, K), gy[t::T].cpu().double(), padding=pad)
                            for t in range(T)], 0)
        werr = ((gw.cpu().double() - wref).abs().max() / wref.abs().max()).item()
        worst = max(worst, err, gerr, werr)
        print(json.dumps({"check": "K%d %d->%d @%dx%d pad %d T%d N%d" % (K, ci, co, H, W, pad, T, N), "fwd_err": err, "cpu_f32_conv_err": err32,
                          "dgrad_err": gerr, "wgrad_err": werr, "ok": bool(err < 3e-6 and gerr < 3e-6 and werr < 3e-6)}), flush=True)
    print(json.dumps({"worst_rel_err": worst}), flush=True)
    return worst


def time_layers(iters, miopen=True):
    dev = torch.device("cuda")
    for (K, ci, co, H, W, T, N, label) in TIME:
        pad = K // 2
        x = torch.randn(N, ci, H, W, device=dev)
        gy = torch.randn(N, co, H, W, device=dev)
        w = torch.randn(T, co, ci, K, K, device=dev) / (K * ci ** 0.5)
        b = torch.randn(T, co, device=dev)
        gflop = 2.0 * K * K * ci * co * H * W * N / 1e9
        row = {"layer": "%dx%d %d->%d @%dx%d T=%d N=%d (%s)" % (K, K, ci, co, H, W, T, N, label), "gflop": round(gflop, 2)}
        pf, pb = hip_ops.convk_filters(w, True, True)
        row["pack_both"] = timeit(lambda: hip_ops.convk_filters(w, True, True), iters)
        row["convk_fwd"] = timeit(lambda: hip_ops.convk_tasks_pre(x, pf, T, ci, co, K, b, 0, 0.0, pad), iters)
        row["convk_dgrad"] = timeit(lambda: hip_ops.convk_tasks_pre(gy, pb, T, ci, co, K, None, 1, 1.0, pad), iters)
        row["convk_fwd_precise"] = timeit(lambda: hip_ops.convk_tasks_pre(x, pf, T, ci, co, K, b, 0, 0.0, pad, True), iters)
        row["convk_wgrad_precise"] = timeit(lambda: hip_ops.convk_wgrad_tasks(x, gy, T, K, pad, True), iters)
        row["convk_wgrad"] = timeit(lambda: hip_ops.convk_wgrad_tasks(x, gy, T, K, pad), iters)
        if K == 3:
            row["wino_wgrad"] = timeit(lambda: hip_ops.conv3x3_wgrad_tasks(x, gy, T, pad), iters)
            uf, ub = hip_ops.conv3x3_filters(w, True, True)
            row["wino_filters"] = timeit(lambda: hip_ops.conv3x3_filters(w, True, True), iters)
            row["wino_fwd"] = timeit(lambda: hip_ops.conv3x3_tasks_pre(x, uf, T, ci, co, b, 0, 0.0, pad), iters)
            row["wino_dgrad"] = timeit(lambda: hip_ops.conv3x3_tasks_pre(gy, ub, T, ci, co, None, 1, 1.0, pad), iters)
        if miopen:
            xs = [x[t::T].contiguous() for t in range(T)]
            gs = [gy[t::T].contiguous() for t in range(T)]
            cb = lambda g_, x_, w_, mask: torch.ops.aten.convolution_backward(g_, x_, w_, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1, mask)
            row["mi_fwd"] = timeit(lambda: [F.conv2d(xs[t], w[t], b[t], padding=pad) for t in range(T)], iters)
            row["mi_dgrad"] = timeit(lambda: [cb(gs[t], xs[t], w[t], [True, False, False]) for t in range(T)], iters)
            row["mi_wgrad"] = timeit(lambda: [cb(gs[t], xs[t], w[t], [False, True, False]) for t in range(T)], iters)
        for k, v in list(row.items()):
            if isinstance(v, float) and k != "gflop":
                row[k] = round(v, 1)
        row["convk_fwd_TF"] = round(gflop / row["convk_fwd"] * 1e3, 1)
        row["convk_dgrad_TF"] = round(gflop / row["convk_dgrad"] * 1e3, 1)
        row["convk_wgrad_TF"] = round(gflop / row["convk_wgrad"] * 1e3, 1)
        if K == 3:
            row["wino_fwd_TF"] = round(gflop / row["wino_fwd"] * 1e3, 1)
        if miopen:
            row["mi_fwd_TF"] = round(gflop / row["mi_fwd"] * 1e3, 1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--no-miopen", action="store_true")
    o = ap.parse_args()
    if o.check or not o.time:
        check()
    if o.time:
        time_layers(o.iters, not o.no_miopen)
