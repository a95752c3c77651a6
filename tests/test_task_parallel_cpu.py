"""CPU, world_size 2, gloo: the task-parallel meta-batch gives the sequential loop's outer gradients.

The real plugins need the GPU (HIP ops), so the N>1 HOST path -- round-robin task sharding, local
loss scaled by the GLOBAL batch, one flat-bucket all-reduce, identical optimizer step on every rank,
logging reduction -- is exercised with a small conv plugin and an oracle-backed inner rule on CPU.
"""
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from meta_interpolation_amd.task_parallel import TaskParallel
from meta_interpolation_amd import synthetic
from tests.helpers import build_toy_system

B, H, W = 5, 16, 24   # 5 tasks over 2 ranks: uneven shards (3 + 2)


def _grads_after_iteration(system, frames, msl):
    captured = {}

    def step(*a, **k):
        captured.update({n: p.grad.detach().clone() for n, p in system.named_parameters() if p.grad is not None})
    system.optimizer.step = step
    losses, preds, _ = system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=False)
    return captured, losses, preds


def _worker(rank, world, port, outdir, msl):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        tp = TaskParallel()
        assert tp.active and tp.world == world and tp.local_tasks(B) == list(range(rank, B, world))
        system = build_toy_system(task_parallel=tp, batch=B, msl=msl)
        frames = synthetic.septuplet_batch(B, H, W)
        grads, losses, preds = _grads_after_iteration(system, frames, msl)
        torch.save({'grads': grads, 'loss_global': losses['loss_global'],
                    'local_preds': [i for i, p in enumerate(preds) if torch.is_tensor(p)]},
                   os.path.join(outdir, "rank%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("msl", [False, True])
def test_sharded_outer_gradients_equal_sequential(msl):
    torch.set_num_threads(2)
    seq = build_toy_system(batch=B, msl=msl)
    want, losses, _ = _grads_after_iteration(seq, synthetic.septuplet_batch(B, H, W), msl)
    assert len(want) > 0 and any(k.startswith('inner_loop_optimizer') for k in want)
    with tempfile.TemporaryDirectory() as d:
        port = 29500 + (os.getpid() % 2000) + (7 if msl else 0)
        mp.spawn(_worker, args=(2, port, d, msl), nprocs=2, join=True)
        ranks = [torch.load(os.path.join(d, "rank%d.pt" % r), weights_only=False) for r in range(2)]
    for r in ranks:
        assert set(r['grads']) == set(want)
        for k, v in want.items():
            assert torch.allclose(r['grads'][k], v, rtol=1e-5, atol=1e-7), k
        assert abs(r['loss_global'] - losses['loss'].item()) < 1e-6
    # replicas end with bit-identical gradients (-> identical optimizer steps, no broadcast needed)
    for k in want:
        assert torch.equal(ranks[0]['grads'][k], ranks[1]['grads'][k])
    assert ranks[0]['local_preds'] == [0, 2, 4] and ranks[1]['local_preds'] == [1, 3]


def test_single_process_task_parallel_is_a_noop():
    tp = TaskParallel()
    assert not tp.active and tp.local_tasks(3) == [0, 1, 2]
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    tp.allreduce_gradients([p])
    assert torch.equal(p.grad, torch.full((3,), 2.0))


# ---------------------------------------------------------------------------------------------
# the dataset provider decodes only the local tasks of a training meta-batch
# ---------------------------------------------------------------------------------------------
def _data_worker(rank, world, port, outdir, root):
    import random
    import types
    from meta_interpolation_amd import data
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        args = types.SimpleNamespace(data_root=root, batch_size=3, val_batch_size=1, test_batch_size=1, mode='train', model='sepconv',
                                     num_gpu=0, num_workers=2, random_seed=5, dataset='vimeo90k', synthetic=False)
        prov = data.MetaLearningSystemDataLoader(args)
        loads = []
        orig = prov.dataset.load
        prov.dataset.load = lambda plan: (loads.append(plan[0][0]), orig(plan))[1]
        random.seed(9)
        train = [(im, me) for im, me in prov.get_train_batches()]
        n_train_loads = len(loads)
        val = [(im, me) for im, me in prov.get_val_batches()]
        torch.save({'train': train, 'val': val, 'n_train_loads': n_train_loads}, os.path.join(outdir, "data%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_provider_decodes_only_local_tasks_and_matches_the_single_process_batches(tmp_path):
    import random
    import types
    from meta_interpolation_amd import data
    root = synthetic.write_fake_vimeo(str(tmp_path / "vimeo"), n_train=6, n_test=2, height=260, width=264)
    args = types.SimpleNamespace(data_root=root, batch_size=3, val_batch_size=1, test_batch_size=1, mode='train', model='sepconv',
                                 num_gpu=0, num_workers=2, random_seed=5, dataset='vimeo90k', synthetic=False)
    random.seed(9)
    full = [(im, me) for im, me in data.MetaLearningSystemDataLoader(args).get_train_batches()]
    with tempfile.TemporaryDirectory() as d:
        port = 29500 + (os.getpid() % 2000) + 21
        mp.spawn(_data_worker, args=(2, port, d, root), nprocs=2, join=True)
        ranks = [torch.load(os.path.join(d, "data%d.pt" % r), weights_only=False) for r in range(2)]
    assert ranks[0]['n_train_loads'] + ranks[1]['n_train_loads'] == 6        # every training item decoded exactly once
    for b, (images, meta) in enumerate(full):
        for r in range(2):
            got_images, got_meta = ranks[r]['train'][b]
            assert got_meta == meta                                              # same crops / flips / paths on every rank
            for t in range(images[0].shape[0]):
                for f in range(7):
                    if t % 2 == r:
                        assert torch.equal(got_images[f][t], images[f][t])       # local task: the real frames
                    else:
                        assert float(got_images[f][t].abs().max()) == 0.0        # someone else's task: never decoded
    assert len(ranks[0]['val']) == len(ranks[1]['val']) == 2                     # validation runs on every rank
    assert all(torch.equal(a[0][3], b[0][3]) for a, b in zip(ranks[0]['val'], ranks[1]['val']))


# ---------------------------------------------------------------------------------------------
# ExperimentBuilder with two ranks: training shards the meta-batch, the end-of-epoch validation sweep does not
# (val_batch_size = 1 < world: a sharded sweep would leave rank 1 without a task, an all-reduce of mismatched length and an
# empty prediction list -- round-1 advisor finding)
# ---------------------------------------------------------------------------------------------
def _eb_worker(rank, world, port, outdir):
    from meta_interpolation_amd.data import SyntheticSeptupletLoader
    from meta_interpolation_amd.experiment_builder import ExperimentBuilder
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    os.chdir(outdir)
    try:
        system = build_toy_system(task_parallel=TaskParallel(), batch=1, steps=1)      # batch 1 < world: rank 1 trains on nothing
        args = system.args
        args.synthetic, args.total_iter_per_epoch, args.max_epoch, args.exp_name, args.log_iter = True, 2, 1, 'toy2', 1
        provider = lambda args, current_iter=0: SyntheticSeptupletLoader(args, current_iter, height=16, width=24,
                                                                         length={'train': 4, 'val': 2, 'test': 1})
        eb = ExperimentBuilder(args, provider, system)
        seen = []
        orig = system.run_validation_iter
        system.run_validation_iter = lambda data_batch: (lambda r: (seen.append((float(r[0]['loss']), r[1][0].clone())), r)[1])(orig(data_batch))
        eb.run_experiment()
        torch.save({'state': {k: v.clone() for k, v in system.state_dict().items()}, 'val': seen, 'epoch': eb.epoch,
                    'best': eb.best_PSNR}, os.path.join(outdir, "eb%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_experiment_builder_trains_and_validates_with_two_ranks(tmp_path):
    port = 29500 + (os.getpid() % 2000) + 33
    mp.spawn(_eb_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(str(tmp_path / ("eb%d.pt" % r)), weights_only=False) for r in range(2))
    assert r0['epoch'] == r1['epoch'] == 1 and len(r0['val']) == len(r1['val']) == 2
    for (la, pa), (lb, pb) in zip(r0['val'], r1['val']):          # every rank evaluated every validation item, same numbers
        assert la == lb and torch.equal(pa, pb)
    assert r0['best'] == r1['best']
    for k, v in r0['state'].items():                                # replicas identical after training + scheduler step
        assert torch.equal(v, r1['state'][k]), k
    assert os.path.exists(str(tmp_path / 'checkpoint' / 'toy2' / 'checkpoint.pth'))


# ---------------------------------------------------------------------------------------------
# bench.py's multi-process host path exactly as the driver launches it (torch.distributed.run, one rank per device),
# on gloo with the toy CPU plugin: rendezvous from the environment, device pinning, task sharding, barrier + max-over-ranks
# timing, ONE JSON line from rank 0
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("task_batch", [0, 2])
def test_bench_runs_under_torch_distributed_run_with_two_ranks(task_batch):
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29500 + (os.getpid() % 2000) + 55 + task_batch
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HIP_VISIBLE_DEVICES"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", "toy_cpu", "--task-batch", str(task_batch)]
    out = subprocess.run(cmd, cwd=repo, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                       # rank 0 prints, rank 1 stays silent
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["global_meta_batch"] == 6 and line["config"]["mode"]["task_batch"] == task_batch
    assert line["value"] > 0 and abs(line["value"] - 6 * 2 * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]
    # the self-validation block of a multi-rank run: backend, world size, ONE collective per meta-iteration, replicas checked after
    # the timed region (an 8-GPU line must carry backend "nccl (RCCL)", rccl_ranks 8 and the all-reduce's own HIP-event time)
    mg = line["multi_gpu"]
    assert mg["backend"] == "gloo" and mg["rccl_ranks"] == 0 and mg["world_size"] == 2
    assert mg["collectives_per_meta_iteration"] == 1 and mg["replicas_bit_identical_after_timed_region"] is True
    assert mg["allreduce"] is None                                   # HIP events exist on the GPU path only
    assert abs(mg["outer_tasks_per_sec_weak"] - line["config"]["outer_tasks_per_sec"]) < 1e-9


# ---------------------------------------------------------------------------------------------
# wider worlds (the 8-GPU node of BASELINE configs 4 / 5): 4 and 8 ranks, 5 tasks -> uneven shards, and with 8 ranks three of
# them adapt NOTHING (they still take part in the all-reduce and must end with the same gradients as everybody else)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("world", [4, 8])
def test_sharded_outer_gradients_equal_sequential_on_wider_worlds(world):
    torch.set_num_threads(2)
    seq = build_toy_system(batch=B, msl=False)
    want, losses, _ = _grads_after_iteration(seq, synthetic.septuplet_batch(B, H, W), False)
    with tempfile.TemporaryDirectory() as d:
        port = 29500 + (os.getpid() % 2000) + 70 + world
        mp.spawn(_worker, args=(world, port, d, False), nprocs=world, join=True)
        ranks = [torch.load(os.path.join(d, "rank%d.pt" % r), weights_only=False) for r in range(world)]
    for r, res in enumerate(ranks):
        assert set(res['grads']) == set(want), r
        for k, v in want.items():
            assert torch.allclose(res['grads'][k], v, rtol=1e-5, atol=1e-7), (r, k)
            assert torch.equal(res['grads'][k], ranks[0]['grads'][k]), (r, k)       # bit-identical replicas
        assert abs(res['loss_global'] - losses['loss'].item()) < 1e-6
        assert res['local_preds'] == list(range(r, B, world))


def _bucket_worker(rank, world, port, outdir):
    """The flat gradient bucket on its own, with the parameter mix of an L2F system: backbone tensors every non-empty rank
    reaches, 'attenuator' tensors only SOME ranks reach this iteration, and a tensor no rank reaches (must stay None so that
    the optimizer skips it: no weight decay, no moment update)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tp = TaskParallel()
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.randn(s)) for s in ((7, 3), (5,), (4, 4), (1,), (6,))]     # backbone x2, attenuator W, gamma_mult, unused
        out = []
        for it in range(2):                    # the second iteration reuses the bucket and the cached presence flags
            views = tp.prepare_gradients(params)
            assert all(p.grad is views[p] for p in params)
            g = torch.Generator().manual_seed(100 * it + rank)
            loss = 0
            if rank % 2 == 0 or rank == world - 1:                 # "has local tasks"
                loss = loss + (params[0] * torch.randn(7, 3, generator=g)).sum() + (params[1] * torch.randn(5, generator=g)).sum()
            if rank == 1 or (it == 1 and rank == 2):               # the L2F tensors: reached by a changing subset of ranks
                loss = loss + (params[2] * torch.randn(4, 4, generator=g)).sum() + (params[3] * 3.0).sum()
            if torch.is_tensor(loss):
                loss.backward()
            tp.allreduce_gradients(params)
            out.append([None if p.grad is None else p.grad.detach().clone() for p in params])
        torch.save(out, os.path.join(outdir, "bucket%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_gradient_bucket_with_partly_reached_parameters(world):
    with tempfile.TemporaryDirectory() as d:
        port = 29500 + (os.getpid() % 2000) + 90 + world
        mp.spawn(_bucket_worker, args=(world, port, d), nprocs=world, join=True)
        ranks = [torch.load(os.path.join(d, "bucket%d.pt" % r), weights_only=False) for r in range(world)]
    for it in range(2):
        want = [torch.zeros(s) for s in ((7, 3), (5,), (4, 4), (1,))]
        for r in range(world):
            g = torch.Generator().manual_seed(100 * it + r)
            if r % 2 == 0 or r == world - 1:
                want[0] += torch.randn(7, 3, generator=g)
                want[1] += torch.randn(5, generator=g)
            if r == 1 or (it == 1 and r == 2):
                want[2] += torch.randn(4, 4, generator=g)
                want[3] += 3.0
        for r in range(world):
            got = ranks[r][it]
            assert got[4] is None                                     # reached by nobody: stays None on every rank
            for a, b in zip(got[:4], want):
                assert a is not None and torch.allclose(a, b, rtol=1e-6, atol=1e-6)
            for a, b in zip(got[:4], ranks[0][it][:4]):
                assert torch.equal(a, b)                              # identical on every rank


# ---------------------------------------------------------------------------------------------
# the REAL SepConv parameter set through the bucket once (world 2, gloo): 21.7 M parameters + learning-rate tables = the 86.7 MB
# message of BASELINE configs 2 / 4 -- layout, views, presence flags and the replica check at the size the 8-GPU run moves
# ---------------------------------------------------------------------------------------------
def _sepconv_bucket_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(4)
    try:
        from meta_interpolation_amd.config import default_args
        from meta_interpolation_amd.meta_learning_system import MODEL_REGISTRY, SceneAdaptiveInterpolation
        args = default_args(model='sepconv', num_gpu=0, batch_size=4, number_of_training_steps_per_iter=5,
                            number_of_evaluation_steps_per_iter=5, optimizer='SGD', loss='1*L1', inner_lr=1e-5)
        args.cuda = False
        net = MODEL_REGISTRY['sepconv'](args, False)
        synthetic.load_seeded_weights(net, 'sepconv')
        tp = TaskParallel()
        system = SceneAdaptiveInterpolation(args, net=net, task_parallel=tp)
        params = [p for p in system.parameters() if p.requires_grad]
        views = tp.prepare_gradients(params)
        nbytes = tp._bucket.numel() * 4
        for i, p in enumerate(params):                 # "the backward pass": gradient of rank r = (r + 1) * (1 + i % 3) everywhere
            views[p].fill_(float((rank + 1) * (1 + i % 3)))
        tp.mark_touched(params)
        tp.allreduce_gradients(params)
        ok = all(p.grad is views[p] and bool((p.grad == float(3 * (1 + i % 3))).all()) for i, p in enumerate(params))
        same = tp.replicas_identical([p.grad for p in params])
        with torch.no_grad():
            if rank == 1:
                params[5].view(-1)[0] += 1e-3          # one diverged element on one rank must be caught
        diverged = tp.replicas_identical(params)
        torch.save(dict(ok=ok, same=same, diverged_detected=not diverged, nbytes=nbytes, n=len(params),
                        numel=sum(p.numel() for p in params)), os.path.join(outdir, "sep%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_real_sepconv_parameter_set_through_the_bucket_with_two_ranks():
    with tempfile.TemporaryDirectory() as d:
        port = 29500 + (os.getpid() % 2000) + 130
        mp.spawn(_sepconv_bucket_worker, args=(2, port, d), nprocs=2, join=True)
        res = [torch.load(os.path.join(d, "sep%d.pt" % r), weights_only=False) for r in range(2)]
    for r in res:
        assert r['ok'] and r['same'] and r['diverged_detected']
        assert r['nbytes'] == 4 * (r['numel'] + r['n'])
        assert 80e6 < r['nbytes'] < 95e6, r['nbytes']          # the 86.7 MB message of DESIGN.md section 6
