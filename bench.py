"""bench.py -- inner-loop steps/sec of the MAML adaptation path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): SepConv, meta-batch 4 tasks per GPU, 5 inner steps, synthetic
256x448x3 septuplets, LSLR + SGD inner rule, L1 loss, outer Adam -- seeded random-init weights.
One "step" of this script = one run_train_iter over the rank's meta-batch (4 tasks x 5 inner steps =
20 inner-loop steps, + target passes, outer backward, all-reduce of outer grads, outer Adam).
`value` = inner-loop steps/sec summed over all ranks (weak scaling: 4 tasks per GPU).

The line also carries
  roofline     : the dominant custom kernel (sepconv backward, gV+gH) -- algorithmic bytes per launch
                 (165.72 MB at B=1,C=3,K=51,384x512: SURVEY.md 8d) / mean launch time from HIP events
                 recorded on the launch stream INSIDE the timed region, against the 8 TB/s HBM peak;
  cpu_baseline : the CPU oracle (oracle/meta.py, the restatement pinned to the reference) timed on the
                 host cores for a bounded sample of the same workload (1 task x 1 inner step at 256x448,
                 incl. target pass and outer backward).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# MIOpen records solver choices per problem shape in a user find-db (~/.config/miopen) and re-uses them in later
# processes - including choices made under torch.backends.cudnn.deterministic=True by the parity tests, which are
# several times slower (measured: 86 -> 15 steps/s when bench.py ran after the test suite on the same box).  The
# benchmark therefore always starts from an empty, private find-db: the numbers are those of a fresh box.
import tempfile  # noqa: E402
os.environ['MIOPEN_USER_DB_PATH'] = tempfile.mkdtemp(prefix='savfi_bench_miopen_')

import torch  # noqa: E402

WORKLOADS = {
    # name: (model, H, W, tasks/GPU, inner steps, overrides)
    'c2_sepconv_256x448_b4_s5': ('sepconv', 256, 448, 4, 5, dict(optimizer='SGD', loss='1*L1', inner_lr=1e-5)),
    'c3_voxelflow_metasgd_256x256_b8_s5': ('voxelflow', 256, 256, 8, 5,
                                           dict(optimizer='Adamax', metasgd=True, loss='1*MSE', inner_lr=1e-5)),
    'c1_cain_64x64_b1_s1': ('cain', 64, 64, 1, 1, dict(optimizer='SGD', loss='1*L1', inner_lr=1e-5)),
    # same launch sequence as C2 on tiny frames: wall time ~= the host-side floor of one C2 meta-iteration
    'c2_host_floor_64x64_b4_s5': ('sepconv', 64, 64, 4, 5, dict(optimizer='SGD', loss='1*L1', inner_lr=1e-5)),
    # SURVEY 8(f) rank 4 plugins (no BASELINE.json config names them: extra lines, same metric)
    'rrin_256x448_b4_s5': ('rrin', 256, 448, 4, 5, dict(optimizer='SGD', loss='1*L1', inner_lr=1e-5)),
    'superslomo_256x448_b4_s5': ('superslomo', 256, 448, 4, 5, dict(optimizer='SGD', loss='1*L1', inner_lr=1e-5)),
    'c5_cain_l2f_720p_b1_s1': ('cain', 720, 1280, 1, 1, dict(optimizer='SGD', loss='1*L1', inner_lr=1e-5, attenuate=True)),
}

def cpu_baseline(model, H, W, overrides):
    """Oracle (CPU restatement) on a bounded sample: 1 task x 1 inner step, full resolution."""
    from meta_interpolation_amd import synthetic
    from oracle import meta, rules
    from tests.helpers import oracle_base
    # N=1 convolutions do not scale past a few dozen threads (256 threads: 10x slower than 32)
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    base = oracle_base(model)
    frames = synthetic.septuplet_batch(1, H, W, model=model)
    kind = 'metasgd' if overrides.get('metasgd') else 'lslr'
    names_w = {n: base[n] for n in meta.inner_param_names([(n, p) for n, p in base.items() if p.is_floating_point()])}
    n_steps = 3
    lrs = rules.init_lrs(kind, names_w, overrides['inner_lr'], num_steps=n_steps)
    t0 = time.perf_counter()
    res = meta.run_iteration(model, base, frames, rule=kind, optimizer=overrides['optimizer'], lrs=lrs,
                             num_steps=n_steps, loss=overrides['loss'].split('*')[1], training=True)
    res['loss'].backward()
    dt = time.perf_counter() - t0
    return {"value": n_steps / dt, "unit": "inner-loop steps/sec", "cores": cores, "kind": "port",
            "sample": "1 task x %d inner steps (each: 2 support fwd+bwd + update) + target pass + outer backward "
                      "at %dx%d, %s, wall %.1f s on %d threads" % (n_steps, H, W, model, dt, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='c2_sepconv_256x448_b4_s5', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timer', action='store_true')
    ap.add_argument('--fuse-conv-act', type=int, default=1)
    ap.add_argument('--graph-inner-loop', type=int, default=0)
    ap.add_argument('--sepconv-window', type=int, default=1)
    ap.add_argument('--task-streams', type=int, default=1, help='tasks adapted concurrently (threads + HIP streams)')
    ap.add_argument('--wgrad-overlap', type=int, default=0, help='weight gradients of support passes on a side stream')
    ap.add_argument('--fast-path', type=int, default=1,
                    help='after the main measurement also time --graph-inner-loop 1 --task-streams 2 (reported as fast_path)')
    opt = ap.parse_args()

    from meta_interpolation_amd import _hip, synthetic, task_parallel
    from meta_interpolation_amd.config import default_args
    from meta_interpolation_amd.meta_learning_system import MODEL_REGISTRY, SceneAdaptiveInterpolation

    if os.environ.get('SAVFI_MIOPEN_FIND'):     # experiment: let MIOpen benchmark its solvers per conv shape
        torch.backends.cudnn.benchmark = True
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    rank, world, local_rank = task_parallel.init_from_env()
    if world != opt.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (opt.gpus, world))
    dev = torch.device('cuda', torch.cuda.current_device())

    model, H, W, tasks, S, over = WORKLOADS[opt.workload]
    args = default_args(model=model, num_gpu=1, batch_size=tasks * world,
                        number_of_training_steps_per_iter=S, number_of_evaluation_steps_per_iter=S,
                        fuse_conv_act=opt.fuse_conv_act, graph_inner_loop=opt.graph_inner_loop,
                        sepconv_window=opt.sepconv_window, task_streams=opt.task_streams, wgrad_overlap=opt.wgrad_overlap, **over)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):      # the ONE line on stdout is the JSON result
        net = MODEL_REGISTRY[model](args, False)
    synthetic.load_seeded_weights(net, model)          # identical theta on every rank, no broadcast
    with contextlib.redirect_stdout(sys.stderr):
        system = SceneAdaptiveInterpolation(args, net=net.to(dev))
    if args.attenuate:   # L2F: non-trivial seeded attenuator (gamma_mult = 0 would make it a no-op)
        sd, gm = synthetic.seeded_attenuator_state(len(system.inner_loop_optimizer.names_learning_rates_dict))
        system.attenuator.load_state_dict(sd)
        with torch.no_grad():
            system.gamma_mult.copy_(gm)
    tp = system.task_parallel

    # the global meta-batch has tasks*world tasks; rank r adapts tasks {t : t mod world == r}.
    # Build only the local ones (others are placeholders that are never touched).
    frames = synthetic.septuplet_batch(tasks * world, H, W, model=model)
    frames = [f.to(dev) for f in frames]                # inputs resident in HBM before the timed region

    def one_iter(it):
        system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=False)

    for i in range(opt.warmup):
        one_iter(i)

    # SURVEY 8(d) second figure: the step bodies alone (2 support passes -> grad -> update), by HIP events around each
    # body on the compute stream.  Eager loop only: hipGraph replays have no per-step host hooks.
    bodies = []
    if not opt.graph_inner_loop and not opt.no_kernel_timer:
        orig_loss, orig_update = system._support_loss, system.apply_inner_loop_update

        import threading
        open_body = threading.local()          # concurrent tasks (--task-streams): one open body per thread / stream

        def support_loss(*a, **k):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            open_body.pair = [ev, None]
            bodies.append(open_body.pair)
            return orig_loss(*a, **k)

        def inner_update(*a, **k):
            out = orig_update(*a, **k)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            pair = getattr(open_body, 'pair', None)
            if pair is not None and pair[1] is None:
                pair[1] = ev
            return out
        system._support_loss, system.apply_inner_loop_update = support_loss, inner_update
    timer = None
    if not opt.no_kernel_timer and model == 'sepconv':
        timer = _hip.KernelTimer(only='sepconv')   # HIP events around the custom sepconv launches only
        _hip.TIMER = timer
    tp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(opt.steps):
        one_iter(i)
    torch.cuda.synchronize()
    tp.barrier()
    elapsed = time.perf_counter() - t0
    _hip.TIMER = None
    if tp.active:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    inner_steps = tasks * world * S * opt.steps
    line = {
        "metric": "inner-loop steps/sec", "value": inner_steps / elapsed, "unit": "inner-loop steps/sec",
        "n_gpus": world, "steps": opt.steps, "warmup": opt.warmup, "ms_per_step": 1e3 * elapsed / opt.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": opt.workload, "plugin": model, "tasks_per_gpu": tasks, "global_meta_batch": tasks * world,
                   "inner_steps": S, "frame": "%dx%dx3" % (H, W), "inner_rule": ("metasgd" if over.get('metasgd') else "lslr")
                   + "+" + over['optimizer'], "parallelism": "task-parallel x%d, 1 all-reduce of outer grads" % world, "task_streams": opt.task_streams, "wgrad_overlap": opt.wgrad_overlap,
                   "outer_tasks_per_sec": tasks * world * opt.steps / elapsed},
    }
    # Second measurement, same workload, same K: the hipGraph-captured inner loop replayed on two task streams.  Reported
    # beside `value`, not as `value`: graph replays carry no per-launch events, and kernels that share the GPU with
    # another stream cannot be priced against a roofline in place.  Parity of this mode: tests/test_system_gpu.py
    # (test_graph_replays_on_two_task_streams_match_reference_fixture).
    fast = None
    from meta_interpolation_amd import graph_inner_loop as _gil
    # single-process runs only: a rank that failed inside this optional block would leave the others waiting in a collective
    if opt.fast_path and world == 1 and not opt.graph_inner_loop and opt.task_streams <= 1:
        saved = (system.args.graph_inner_loop, system.args.task_streams)
        system.args.graph_inner_loop, system.args.task_streams = 1, 2
        try:
            if _gil.supported(system, bool(getattr(system.args, 'second_order', False))):
                for i in range(max(opt.warmup, 2)):
                    one_iter(i)                      # captures one graph set per stream
                tp.barrier()
                torch.cuda.synchronize()
                f0 = time.perf_counter()
                for i in range(opt.steps):
                    one_iter(i)
                torch.cuda.synchronize()
                tp.barrier()
                fel = time.perf_counter() - f0
                if tp.active:
                    import torch.distributed as dist
                    ft = torch.tensor([fel], dtype=torch.float64, device=dev)
                    dist.all_reduce(ft, op=dist.ReduceOp.MAX)
                    fel = float(ft.item())
                fast = {"value": inner_steps / fel, "unit": "inner-loop steps/sec", "ms_per_step": 1e3 * fel / opt.steps,
                        "mode": "--graph-inner-loop 1 --task-streams 2",
                        "note": "same workload and step count, measured after the main region; first-order inner loop replayed from "
                                "hipGraphs on two task streams, outer gradients assembled by hand"}
        except Exception as e:       # optional figure: never lose the main line over it
            fast = {"error": "%s: %s" % (type(e).__name__, str(e)[:200]), "mode": "--graph-inner-loop 1 --task-streams 2"}
        finally:
            system.args.graph_inner_loop, system.args.task_streams = saved
    if fast is not None:
        line["fast_path"] = fast
    done = [(a, b) for a, b in bodies if b is not None]
    if done:
        body_ms = sum(a.elapsed_time(b) for a, b in done)
        if opt.task_streams <= 1:      # bodies of concurrent tasks overlap in time: their sum is not a share of the wall clock
            line["config"]["step_bodies_only_steps_per_sec"] = len(done) / (body_ms * 1e-3) * world
            line["config"]["step_bodies_share_of_iteration"] = body_ms * 1e-3 / elapsed
    if rank == 0:
        if timer is not None:
            summ = timer.summary()
            line["kernels"] = summ
            k = summ.get("sepconv_bwd")
            if k:
                # the op runs on the frame window (H x W) or, with --sepconv-window 0, on the reference's padded canvas
                oh, ow = (H, W) if opt.sepconv_window else net.padded_size(H, W)
                traffic, tnote = None, "no committed PMC measurement found"
                tpath = os.path.join(REPO, "profiles", "r01_hbm_traffic_sepconv.json")
                if os.path.exists(tpath):
                    # HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
                    # (tools/hbm_traffic.py; FETCH_SIZE x2 per the gfx950 calibration), scaled to this run's
                    # B=1 / B=2 launch mix through the measured traffic / algorithmic ratio of each shape
                    tk = json.load(open(tpath))["kernels"]
                    keys = ["sepconv_bwd_B%d_%dx%d" % (b, oh, ow) for b in (1, 2)]
                    if all(kk in tk for kk in keys):
                        ratio = 0.5 * sum(tk[kk]["traffic_over_algorithmic"] for kk in keys)
                        traffic = ratio * k["algorithmic_bytes"] / k["launches"]
                        tnote = ("PMC (FETCH_SIZE*2 + WRITE_SIZE) = %.3f x algorithmic at %dx%d, "
                                 "profiles/r01_hbm_traffic_sepconv.json" % (ratio, oh, ow))
                per_call = 4.0 * (3 * (oh + 50) * (ow + 50) + 4 * 51 * oh * ow + 3 * oh * ow)
                line["roofline"] = {
                    "bound": "hbm", "kernel": "sepconv_bwd_mfma (gV+gH, K=51)",
                    "achieved": k["achieved_GBps"], "peak": 8000.0, "unit": "GB/s",
                    "frac": k["achieved_GBps"] / 8000.0, "traffic": traffic, "traffic_source": tnote,
                    "avg_us_per_launch": k["avg_us"], "launches": k["launches"],
                    "algorithmic_bytes_per_launch": k["algorithmic_bytes"] / k["launches"],
                    "note": "algorithmic bytes = %.2f MB per [1,3,%d,%d] call (x2 for the fused N=2 support "
                            "pair); fp32 issue ceiling of this kernel is ~53%% of HBM peak (SURVEY.md 7)"
                            % (per_call / 1e6, oh, ow)}
        if world == 1 and not opt.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(model, H, W, over)
        print(json.dumps(line), flush=True)
    tp.barrier()


if __name__ == '__main__':
    main()
