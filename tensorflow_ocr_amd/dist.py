"""Data-parallel gradient exchange: the RCCL (or gloo, for CPU tests) all-reduce(mean) that
replaces `average_gradients` (multigpu_train.py:70-85: concat + reduce_mean of the per-tower
gradients) and, with `op="sum"`, `sum_gradients` of train_pixellink.py:179-194.

One process per GPU.  The tower's gradients live in ONE flat f32 buffer (graph.VariableStore);
it is cut into contiguous buckets.  During backward, as soon as every variable of a bucket has its
gradient (Graph.backward's `on_grads_ready` hook) the bucket's all-reduce is issued on a side
stream that waits on an event recorded on the compute stream, so communication overlaps the rest
of backward; `finish()` makes the compute stream wait for all buckets before the optimiser runs.

xGMI sizing: 8 MI355X are fully connected, 7 links x ~153 GB/s per GPU; a ring is
bound by one link, so buckets are kept large (default 32 MB -> 3 buckets for VGG's 82 MB) to
amortise per-collective latency while still letting the fc6/fc7/conv5 gradients (the first to
complete, ~60 % of the bytes) travel under the conv1-conv4 backward.
"""
import torch
import torch.distributed as td


def init_process_group_from_env(backend=None, force=False):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run contract; also what
    `launch.self_launch` sets).  A single tower needs no group; `force=True` (or OCR_FORCE_PG=1)
    creates a ONE-rank group anyway, so the RCCL bucket path can be exercised on a one-GPU box."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    force = force or os.environ.get("OCR_FORCE_PG", "0") == "1"
    if world == 1 and not force:
        return 0, 1, 0
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", rank))
    if world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            from .launch import free_port
            os.environ["MASTER_PORT"] = str(free_port())
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not td.is_initialized():
        td.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def plan_buckets(var_ranges, total, bucket_elems):
    """var_ranges: [(start, stop)] of each variable in the flat buffer, in creation order.
    Returns [(start, stop)] bucket ranges (contiguous, covering [0,total)) with boundaries on
    variable boundaries, filled from the END of the buffer backwards — gradients complete in
    reverse creation order, so the last bucket is ready first; the first bucket (ready last, never
    overlapped) is kept to a quarter of the bucket size."""
    if not var_ranges:
        return [(0, total)]
    bounds = sorted(set([0, total] + [s for s, _ in var_ranges]))
    buckets = []
    idx = len(bounds) - 1            # bounds[idx] == stop of the bucket being built
    while idx > 0:
        j = idx - 1                  # at least one variable per bucket
        while j - 1 >= 0 and bounds[idx] - bounds[j - 1] <= bucket_elems:
            j -= 1
        buckets.append((bounds[j], bounds[idx]))
        idx = j
    buckets = buckets[::-1]
    # the FIRST bucket holds the earliest layers, whose gradients complete last: its all-reduce cannot
    # hide under any backward work, so keep it small (<= a quarter bucket) by splitting off its tail
    s0, e0 = buckets[0]
    if e0 - s0 > bucket_elems // 4:
        inner = [b for b in bounds if s0 < b < e0]
        cut = None
        for b in inner:
            if b - s0 <= bucket_elems // 4:
                cut = b
        if cut is None and inner:
            cut = inner[0]
        if cut is not None:
            buckets = [(s0, cut), (cut, e0)] + buckets[1:]
    return buckets


class GradientAllReduce:
    def __init__(self, store, world_size, bucket_bytes=32 << 20, op="mean", group=None, fold_mean=False,
                 force=False):
        """op: "mean" = `average_gradients` (multigpu_train.py:70-85); "sum" = `sum_gradients`
        (train_pixellink.py:179-194: the caller has already divided its loss by num_clones).
        fold_mean: leave the SUM in the buffer and let the optimiser apply `grad_scale` (= 1/world)
        inside its own pass instead of one more sweep over the gradients.
        force: run the bucket / comm-stream / wait path at world 1 too (a one-rank group must exist).
        `enabled = False` turns every hook into a no-op (bench.py's comm-exposed A/B: the step without
        its exchange)."""
        self.store = store
        self.active = world_size > 1 or force
        self.enabled = True
        self.fold_mean = fold_mean
        self.grad_scale = (1.0 / world_size) if (fold_mean and op == "mean") else 1.0
        self.world = world_size
        self.op = op
        self.group = group
        base = store.flat.data_ptr()
        self.var_range = {}
        ranges = []
        for v in store.trainable():
            off = (v.data.data_ptr() - base) // 4
            self.var_range[v.name] = (off, off + v.size)
            ranges.append((off, off + v.size))
        self.buckets = plan_buckets(ranges, store.flat.numel(), max(bucket_bytes // 4, 1))
        self.bucket_of = {}
        self.need = [0] * len(self.buckets)
        for name, (s, e) in self.var_range.items():
            for bi, (bs, be) in enumerate(self.buckets):
                if bs <= s < be:
                    self.bucket_of[name] = bi
                    self.need[bi] += 1
                    break
        self.cuda = store.flat.is_cuda
        self.comm_stream = torch.cuda.Stream() if self.cuda else None
        self.extra_streams = []       # streams that also produce gradients (side-stream wgrad)
        self.reset()

    def reset(self):
        self.left = list(self.need)
        self.handles = []
        self.fired = [False] * len(self.buckets)

    def bucket_nbytes(self):
        return [(e - s) * 4 for s, e in self.buckets]

    def on_grads_ready(self, variables):
        if not self.active or not self.enabled:
            return
        for v in variables:
            bi = self.bucket_of.get(v.name)
            if bi is None:
                continue
            self.left[bi] -= 1
            if self.left[bi] == 0 and not self.fired[bi]:
                self._fire(bi)

    def _fire(self, bi):
        self.fired[bi] = True
        s, e = self.buckets[bi]
        buf = self.store.flat_grad[s:e]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                for es in self.extra_streams:
                    self.comm_stream.wait_stream(es)
                h = td.all_reduce(buf, op=td.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            h = td.all_reduce(buf, op=td.ReduceOp.SUM, group=self.group, async_op=True)
        self.handles.append((h, buf))

    def finish(self):
        """Fire whatever has not been fired (variables without a gradient this step), wait for all
        buckets, and turn the sum into the tower mean (multigpu_train.py:80-81)."""
        if not self.active or not self.enabled:
            self.reset()
            return
        for bi in range(len(self.buckets)):
            if not self.fired[bi]:
                self._fire(bi)
        for h, buf in self.handles:
            h.wait()          # cuda: makes the current stream wait for the collective
        if self.op == "mean" and not self.fold_mean:
            if self.cuda:
                from . import ops
                ops.scale_(self.store.flat_grad, 1.0 / self.world)
            else:
                self.store.flat_grad.mul_(1.0 / self.world)
        self.reset()


def any_rank(flag, device=None):
    """True on EVERY rank iff `flag` is true on ANY rank (all-reduce MAX of one int).  The stop
    decision of the training loops (`np.isnan(loss)` -> break, multigpu_train.py:175-177) must be
    collective: a rank that leaves the loop alone strands the others in the next all-reduce."""
    if not td.is_initialized() or td.get_world_size() == 1:
        return bool(flag)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if td.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return bool(t.item())
