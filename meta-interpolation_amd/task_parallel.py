"""Task-parallel meta-batch: one process per GPU, tasks sharded round-robin, ONE all-reduce of the
outer gradients per meta-iteration (RCCL over xGMI on the GPU box; gloo in the CPU tests).

The reference has no multi-GPU path at all: tasks of a meta-batch are a
sequential Python loop (meta_learning_system.py:366).  They are independent -- every task starts
from the same theta (:370), owns its rule state (:377) and adds one term to mean_t L_t (:338) -- so
rank r takes tasks {t : t mod G == r}, back-propagates sum_local L_t / B_global, and a single SUM
all-reduce of one flat fp32 bucket (net + lr tables + attenuator grads, ~87 MB for SepConv) restores
exactly the sequential gradient.  Every rank then applies the identical optimizer step, so replicas
stay bit-identical without any parameter broadcast.  xGMI is point-to-point: one large message per
iteration (instead of per-tensor calls) keeps each ring hop a single bandwidth-bound transfer.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT
    (set by torch.distributed.run).  Returns (rank, world, local_rank).  No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and os.environ.get("SAVFI_PIN_DEVICE") == "1" and "HIP_VISIBLE_DEVICES" not in os.environ \
            and "CUDA_VISIBLE_DEVICES" not in os.environ and not torch.cuda.is_initialized():
        # Opt-in (SAVFI_PIN_DEVICE=1): hide every device but this rank's before the HIP runtime comes up.  Not the
        # default: under ROCm, isolating ranks with *_VISIBLE_DEVICES can take peer-to-peer IPC away from RCCL (ranks that cannot
        # see each other's GPU fall back to host staging), and the one all-reduce of a meta-iteration wants xGMI.  One process
        # still owns ONE device either way: everything below runs on `local_rank` only.
        os.environ["HIP_VISIBLE_DEVICES"] = str(local_rank)
    if torch.cuda.is_available():
        pinned = os.environ.get("HIP_VISIBLE_DEVICES", "").strip() == str(local_rank) and torch.cuda.device_count() == 1
        torch.cuda.set_device(0 if pinned else local_rank % max(torch.cuda.device_count(), 1))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # SAVFI_DIST_BACKEND=gloo: the multi-rank GPU path on a box with fewer GPUs than ranks (ranks share device
            # local_rank % device_count; RCCL refuses two ranks on one device, gloo stages CUDA tensors through the host)
            backend = os.environ.get("SAVFI_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")   # "nccl" is RCCL on ROCm
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = torch.device("cuda", torch.cuda.current_device())
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank


class TaskParallel:
    """Sharding + gradient exchange policy.  ``TaskParallel()`` picks up the default process group if
    one is initialised, else behaves as a single rank (every method is then a cheap no-op)."""

    def __init__(self, group=None):
        self.group = group
        if dist.is_available() and dist.is_initialized():
            self.rank = dist.get_rank(group)
            self.world = dist.get_world_size(group)
        else:
            self.rank, self.world = 0, 1
        self._bucket = None     # flat fp32 buffer: every trainable gradient + one presence flag per parameter
        self._bucket_key = None
        self._views = None      # per-parameter views of the bucket (these ARE the .grad tensors during an outer step)
        self._flag_cache = {}   # local presence pattern -> device tensor of flags
        self._touched = set()   # id(param) of the parameters the current backward pass reached
        self._hooked = set()
        self.record_timing = False      # bench.py: HIP events around every gradient all-reduce (the stream the collective is ordered on)
        self._timing = []               # [(start event, end event, bucket bytes)]

    @property
    def active(self):
        return self.world > 1

    def local_tasks(self, num_tasks):
        """Indices of the meta-batch this rank adapts (round-robin: task t -> rank t mod G)."""
        return [t for t in range(num_tasks) if t % self.world == self.rank]

    # -- the one collective of a meta-iteration ---------------------------------------------
    def _ensure_bucket(self, params):
        sizes = [p.numel() for p in params]
        total = sum(sizes)
        dev = params[0].device
        key = (tuple(id(p) for p in params), tuple(sizes), str(dev))
        if self._bucket is None or self._bucket_key != key:
            self._bucket = torch.empty(total + len(params), dtype=torch.float32, device=dev)
            self._bucket_key = key
            self._views = [v.view_as(p) for v, p in zip(self._bucket[:total].split(sizes), params)]
            self._flag_cache = {}
            for p in params:        # which parameters a backward pass reached is recorded on the HOST (no device sync)
                if id(p) not in self._hooked:
                    self._hooked.add(id(p))
                    p.register_post_accumulate_grad_hook(lambda q, _t=self._touched: _t.add(id(q)))
        return self._bucket, self._views

    def prepare_gradients(self, params):
        """Before the outer backward: every .grad becomes a zeroed VIEW of the flat bucket, so autograd accumulates straight
        into the buffer that is all-reduced (no copy into the bucket, none back out).  Returns {param: view} (the graphed
        path writes its hand-assembled gradients there, then calls mark_touched) or None for a single process."""
        if not self.active:
            return None
        params = [p for p in params]
        if not params:
            return None
        bucket, views = self._ensure_bucket(params)
        bucket.zero_()
        self._touched.clear()
        for p, v in zip(params, views):
            p.grad = v
        return dict(zip(params, views))

    def mark_touched(self, params):
        self._touched.update(id(p) for p in params)

    def allreduce_gradients(self, params):
        """SUM-reduce .grad of `params` (fixed order on every rank) through one flat bucket.
        A parameter that received no gradient locally contributes zeros; it ends with a gradient iff
        some rank produced one (presence flags ride at the tail of the same bucket)."""
        if not self.active:
            return
        params = [p for p in params]
        if not params:
            return
        bucket, views = self._ensure_bucket(params)
        total = bucket.numel() - len(params)
        prepared = all(p.grad is v for p, v in zip(params, views))
        if prepared:            # gradients already live in the bucket (prepare_gradients): nothing to copy
            have = [id(p) in self._touched for p in params]
        else:                   # stand-alone use: gather whatever .grad tensors exist
            have = [p.grad is not None for p in params]
            src = [p.grad.reshape(-1) for p, h in zip(params, have) if h]
            dst = [v.reshape(-1) for v, h in zip(views, have) if h]
            if src:
                torch._foreach_copy_(dst, src)
            for v, h in zip(views, have):
                if not h:
                    v.zero_()
        key = tuple(have)
        dev_flags = self._flag_cache.get(key)
        if dev_flags is None:   # one H2D per presence pattern, device-to-device afterwards
            dev_flags = self._flag_cache[key] = torch.tensor([1.0 if h else 0.0 for h in have], dtype=torch.float32).to(bucket.device)
        bucket[total:].copy_(dev_flags, non_blocking=True)
        timed = self.record_timing and bucket.is_cuda
        if timed:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group)
        if timed:
            ev1.record()
            self._timing.append((ev0, ev1, bucket.numel() * bucket.element_size()))
        # A parameter ends with a gradient iff some rank produced one (a None gradient makes the optimizer skip it: no weight
        # decay, no moment update -- zeros would not be the same thing).  When this rank produced every gradient itself the
        # answer is known without looking; otherwise (an empty or partial shard: the rare case) the reduced flags are fetched,
        # every time -- which parameters the OTHER ranks reached can change with the pass (MSL epochs ending, eager vs graphed).
        keep_all = have if all(have) else (bucket[total:].cpu() > 0).tolist()
        for p, v, h, keep in zip(params, views, have, keep_all):
            if not keep:
                p.grad = None
            elif prepared or not h:
                p.grad = v          # the reduced sum, in place in the bucket
            else:
                p.grad.copy_(v)

    def allreduce_stats(self, reset=True):
        """HIP-event time of the recorded gradient all-reduces (call after a device synchronize): launches, mean / max ms, bucket
        bytes, and the bus bandwidth a ring moves for it (2 (G - 1) / G x bytes / time).  None if nothing was recorded."""
        if not self._timing:
            return None
        ms = [a.elapsed_time(b) for a, b, _ in self._timing]
        nbytes = self._timing[-1][2]
        if reset:
            self._timing = []
        mean = sum(ms) / len(ms)
        return {"launches": len(ms), "mean_ms": mean, "max_ms": max(ms), "bucket_bytes": nbytes,
                "algorithm_GBps": nbytes / mean / 1e6 if mean > 0 else None,
                "bus_GBps": 2.0 * (self.world - 1) / self.world * nbytes / mean / 1e6 if mean > 0 else None}

    def replicas_identical(self, tensors):
        """True iff every rank holds bit-identical copies of `tensors` (one all-gather of two float64 checksums per rank: sum and
        sum of squares over everything; identical replicas give identical sums bit for bit)."""
        if not self.active:
            return True
        tensors = [t.detach() for t in tensors]
        dev = tensors[0].device
        acc = torch.zeros(2, dtype=torch.float64, device=dev)
        for t in tensors:
            d = t.double()
            acc[0] += d.sum()
            acc[1] += (d * d).sum()
        if dist.get_backend(self.group) == "nccl" and not acc.is_cuda:
            acc = acc.cuda()
        box = [torch.empty_like(acc) for _ in range(self.world)]
        dist.all_gather(box, acc, group=self.group)
        return all(torch.equal(b.cpu(), box[0].cpu()) for b in box)

    def allreduce_scalars(self, values):
        """SUM a small list/1-D tensor of logging scalars; returns a tensor on the input's device."""
        t = values if torch.is_tensor(values) else torch.tensor(values, dtype=torch.float64)
        if self.active:
            if dist.get_backend(self.group) == "nccl" and not t.is_cuda:
                t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def union_of_keys(self, keys):
        """Sorted union of every rank's string keys (logging only; a tiny object all-gather)."""
        if not self.active:
            return sorted(keys)
        box = [None] * self.world
        dist.all_gather_object(box, list(keys), group=self.group)
        return sorted(set(k for ks in box for k in ks))

    def barrier(self):
        if self.active:
            dist.barrier(group=self.group)
