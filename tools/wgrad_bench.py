"""3x3 weight-gradient micro-benchmark: savfi_conv3x3_wgrad_f32 vs MIOpen through aten::convolution_backward (its layout
transposes and zero fills included), on the SepConv layer shapes.  python tools/wgrad_bench.py [--n 2]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meta_interpolation_amd import hip_ops  # noqa: E402
from tools.conv_bench import EXTRA, LAYERS, timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--set", default="sepconv", choices=["sepconv"] + sorted(EXTRA))
    o = ap.parse_args()
    dev = torch.device("cuda")
    for (ci, co, h, w) in (LAYERS if o.set == "sepconv" else EXTRA[o.set]):
        x = torch.randn(o.n, ci, h, w, device=dev)
        wt = torch.randn(co, ci, 3, 3, device=dev)
        gy = torch.randn(o.n, co, h, w, device=dev)
        ref = torch.ops.aten.convolution_backward(gy, x, wt, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        got = hip_ops.conv3x3_wgrad(x, gy, 1)
        err = float((got - ref).abs().max() / ref.abs().max())
        t_mi = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, wt, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                                  [False, True, False]), o.iters)
        t_my = timeit(lambda: hip_ops.conv3x3_wgrad(x, gy, 1), o.iters)
        gflop = 2.0 * 9 * ci * co * h * w * o.n / 1e9
        print(json.dumps({"layer": "%d->%d @%dx%d N=%d" % (ci, co, h, w, o.n), "gflop": round(gflop, 2), "miopen_us": round(t_mi, 1),
                          "savfi_us": round(t_my, 1), "ratio": round(t_mi / t_my, 2), "savfi_TFLOPs": round(gflop / t_my * 1e3, 1),
                          "rel_err_vs_miopen": err}), flush=True)


if __name__ == "__main__":
    main()
