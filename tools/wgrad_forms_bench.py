"""Weight gradient of the 3x3 layers, three ways: the direct MFMA kernel (savfi_conv3x3_wgrad_tasks_f32), the Winograd form
F(3x3, 2x2) (savfi_conv3x3_wgrad_wino_tasks_f32) and MIOpen (one grouped convolution_backward over the T tasks; plain for T = 1).
First block: accuracy against autograd in float64; second: device time per call on layer shapes of SepConv in lockstep (T = 4 tasks,
n = 2 samples each), its shared-weight layers (N = 8) and CAIN / U-Net shapes.

    python tools/wgrad_forms_bench.py
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meta_interpolation_amd import _hip  # noqa: E402
from tools.conv_bench import timeit  # noqa: E402


def form(entry, x, gz, T, pad):
    lib = _hip.lib()
    N, Ci, H, W = x.shape
    Co = gz.shape[1]
    ws = torch.empty(int(getattr(lib, entry.replace('_f32', '_workspace_floats'))(N, T, Ci, Co, H, W, pad)), device=x.device)
    gw = torch.empty((T, Co, Ci, 3, 3), device=x.device)
    _hip.check(getattr(lib, entry)(x.data_ptr(), gz.data_ptr(), gw.data_ptr(), ws.data_ptr(), N, T, Ci, Co, H, W, pad, _hip.current_stream()), entry)
    return gw


DIRECT, WINO = 'savfi_conv3x3_wgrad_tasks_f32', 'savfi_conv3x3_wgrad_wino_tasks_f32'


def main():
    for (T, N, Ci, Co, H, W, pad) in [(1, 1, 3, 5, 6, 8, 1), (1, 2, 32, 32, 16, 16, 1), (2, 4, 6, 32, 24, 40, 1), (1, 1, 64, 51, 37, 45, 1),
                                      (1, 2, 51, 51, 18, 30, 0), (4, 8, 32, 32, 20, 30, 1)]:
        g = torch.Generator().manual_seed(3)
        x = torch.randn(N, Ci, H, W, generator=g)
        gz = torch.randn(N, Co, H + 2 * pad - 2, W + 2 * pad - 2, generator=g)
        want = []
        for t in range(T):
            w = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
            want.append(torch.autograd.grad(F.conv2d(x[t::T].double(), w, None, padding=pad), w, gz[t::T].double())[0])
        want = torch.stack(want)
        err = lambda e: float(((form(e, x.cuda(), gz.cuda(), T, pad).cpu().double() - want).abs().max() / want.abs().max()))
        print("T=%d N=%d %d->%d @%dx%d pad %d: rel err vs float64  direct %.1e  winograd %.1e" % (T, N, Ci, Co, H, W, pad, err(DIRECT), err(WINO)), flush=True)
    cb = lambda g_, x_, w_, groups: torch.ops.aten.convolution_backward(g_, x_, w_, None, [1, 1], [1, 1], [1, 1], False, [0, 0], groups, [False, True, False])
    shapes = [(4, 2, 6, 32, 384, 512), (4, 2, 32, 32, 384, 512), (4, 2, 32, 64, 192, 256), (4, 2, 64, 64, 192, 256), (4, 2, 128, 128, 96, 128),
              (4, 2, 256, 256, 48, 64), (4, 2, 256, 512, 24, 32), (4, 2, 512, 512, 24, 32), (4, 2, 512, 512, 12, 16), (4, 2, 256, 256, 24, 32),
              (4, 2, 128, 128, 48, 64), (4, 2, 64, 64, 96, 128), (1, 8, 64, 64, 136, 233), (1, 8, 51, 51, 258, 450), (1, 2, 192, 192, 96, 160),
              (1, 2, 128, 128, 64, 112), (1, 1, 64, 64, 192, 256), (1, 2, 32, 32, 256, 448)]
    for (T, n, Ci, Co, H, W) in shapes:
        x = torch.randn(n * T, Ci, H, W, device='cuda')
        gz = torch.randn(n * T, Co, H, W, device='cuda')
        wt = torch.randn(T * Co, Ci, 3, 3, device='cuda')
        a = timeit(lambda: form(DIRECT, x, gz, T, 1), 6)
        b = timeit(lambda: form(WINO, x, gz, T, 1), 6)
        xg, gg = x.view(n, T * Ci, H, W), gz.view(n, T * Co, H, W)
        m = timeit(lambda: cb(gg, xg, wt, T), 3, reps=3) if (H * W < 3100 or T == 1) else float('nan')   # grouped MIOpen on large maps: 3-10 ms
        gf = 18e-9 * Ci * Co * H * W * n * T
        print("%3d->%3d @%dx%d T=%d n=%d (%.1f GFLOP): direct %.1f us  winograd %.1f us (%.0f direct-equivalent TFLOP/s)  MIOpen %.1f us"
              % (Ci, Co, H, W, T, n, gf, a, b, gf / b * 1e3, m), flush=True)


if __name__ == '__main__':
    main()
