"""SAVFI_CONVK_TILE sweep of the direct 3x3 kernel (forward and data gradient) on the batched Subnet shapes of round 4 (the four Subnets as
one launch per layer: N = 32, T = 4 at 137 x 233; layer 1 as one 64 -> 256 convolution) -- one process per setting (the knob is read per
call, but the packed filters differ with nt)."""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [(64, 64, 137, 233, 4, 32), (64, 256, 137, 233, 1, 8), (64, 51, 137, 233, 4, 32)]


def child():
    import torch
    from meta_interpolation_amd import hip_ops
    dev = torch.device("cuda")

    def timeit(fn, iters=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        evs = []
        for _ in range(iters):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); evs.append((a, b))
        torch.cuda.synchronize()
        t = sorted(1e3 * a.elapsed_time(b) for a, b in evs)
        return t[len(t) // 2]
    for (ci, co, H, W, T, N) in SHAPES:
        x = torch.randn(N, ci, H, W, device=dev); gy = torch.randn(N, co, H, W, device=dev)
        w = torch.randn(T, co, ci, 3, 3, device=dev) / (3 * ci ** 0.5)
        pf, pb = hip_ops.convk_filters(w, True, True)
        fl = 2.0 * 9 * ci * co * H * W * N
        tf = timeit(lambda: hip_ops.convk_tasks_pre(x, pf, T, ci, co, 3, None, 0, 0.0, 1))
        tb = timeit(lambda: hip_ops.convk_tasks_pre(gy, pb, T, ci, co, 3, None, 1, 1.0, 1))
        print(json.dumps({"tile": os.path.basename(os.environ.get("SAVFI_HIP_LIB", "heuristic")), "layer": "%d->%d @%dx%d T=%d N=%d" % (ci, co, H, W, T, N),
                          "fwd_us": round(tf, 1), "fwd_TF": round(fl / tf / 1e6), "dgrad_us": round(tb, 1), "dgrad_TF": round(fl / tb / 1e6)}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for tile in (None, "4,1", "4,2", "2,1", "2,2"):
            env = dict(os.environ)
            if tile:
                # a variant per tile: tools/build_variant.sh tile_N_C convk.hip -DSAVFI_CONVK_TILE_NT=N -DSAVFI_CONVK_TILE_CG=C
                env["SAVFI_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "libsavfi_tile_%s.so" % tile.replace(",", "_"))
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True).stdout
            print("".join(l + "\n" for l in out.splitlines() if l.startswith("{")), end="", flush=True)
