"""HIP-event timing of the element-wise glue kernels on SepConv's C2 map shapes (N = 8 support pair x 4 tasks): savfi_bias_act_bwd_f32 in its
three uses (derivative + bias sums, derivative only, bias sums only) and GB/s against the bytes each use has to move."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meta_interpolation_amd import _hip

dev = torch.device("cuda")
lib, st = _hip.lib(), _hip.current_stream()


def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(1e3 * a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2]


for (N, C, H, W) in [(8, 32, 384, 512), (8, 64, 192, 256), (8, 128, 96, 128), (8, 256, 48, 64), (8, 512, 24, 32), (8, 512, 12, 16), (32, 64, 137, 233), (32, 51, 137, 233)]:
    gy = torch.randn(N, C, H, W, device=dev); y = torch.randn(N, C, H, W, device=dev).relu_(); gz = torch.empty_like(gy)
    gb = torch.empty(C, device=dev)
    scratch = torch.empty(int(lib.savfi_bias_act_scratch_floats(N, C, H * W)), device=dev)
    nb = gy.numel() * 4
    full = timeit(lambda: _hip.check(lib.savfi_bias_act_bwd_f32(gy.data_ptr(), y.data_ptr(), gz.data_ptr(), gb.data_ptr(), scratch.data_ptr(), N, C, H * W, 0.0, st), "b"))
    mask = timeit(lambda: _hip.check(lib.savfi_bias_act_bwd_f32(gy.data_ptr(), y.data_ptr(), gz.data_ptr(), None, None, N, C, H * W, 0.0, st), "b"))
    sums = timeit(lambda: _hip.check(lib.savfi_bias_act_bwd_f32(gy.data_ptr(), gy.data_ptr(), None, gb.data_ptr(), scratch.data_ptr(), N, C, H * W, 1.0, st), "b"))
    print(json.dumps({"map": [N, C, H, W], "MB": round(nb / 1e6, 1), "deriv+sums_us": round(full, 1), "deriv+sums_GBps": round(3 * nb / full / 1e3),
                      "deriv_us": round(mask, 1), "deriv_GBps": round(3 * nb / mask / 1e3), "sums_us": round(sums, 1), "sums_GBps": round(nb / sums / 1e3)}), flush=True)
