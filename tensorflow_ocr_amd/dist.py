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


class AbiComm:
    """An RCCL communicator created and used through the C ABI (include/ocr_hip.h: ocr_comm_*,
    ocr_allreduce_bucket).  The 128-byte unique id travels from rank 0 to the others through
    torch.distributed's key-value store (the rendezvous the launcher already set up); at world 1 no
    rendezvous is needed at all.  Must be created with the rank's device current."""

    _serial = 0
    ACK_TIMEOUT_S = 120

    def __init__(self, rank=0, world=1, store=None):
        import ctypes
        from . import _lib as L
        if not L.call_int("ocr_comm_available"):
            raise L.OcrHipError("RCCL is not available: %s" % _last_comm_error())
        ident = ctypes.create_string_buffer(128)
        key = "ocr_comm_id_%d" % AbiComm._serial
        AbiComm._serial += 1
        if world > 1:
            if store is None:
                store = td.distributed_c10d._get_default_store()
            err = None
            try:
                if rank == 0:
                    rc = L._fn("ocr_comm_unique_id", ctypes.c_int)(ident)
                    # a failure here must reach the other ranks too: they would wait for the key, then for this rank
                    # inside ncclCommInitRank
                    store.set(key, ident.raw if rc == 0 else b"ERR")
                    self._check(rc, "ocr_comm_unique_id")
                else:
                    raw = bytes(store.get(key))
                    if raw == b"ERR":
                        raise L.OcrHipError("rank 0 could not create an RCCL unique id")
                    ident = ctypes.create_string_buffer(raw, 128)
            except Exception as e:                           # store timeout, refused id, ...
                err = e
            # second gate: nobody enters ncclCommInitRank — which blocks until ALL ranks arrive — before every
            # rank has said that it holds the id; a rank that failed above says so, and all ranks raise together
            import datetime
            store.set("%s_ack_%d" % (key, rank), b"OK" if err is None else b"ERR")
            acks = ["%s_ack_%d" % (key, r) for r in range(world)]
            try:
                store.wait(acks, datetime.timedelta(seconds=self.ACK_TIMEOUT_S))
                bad = [r for r in range(world) if bytes(store.get(acks[r])) != b"OK"]
            except Exception as e:                           # a rank never answered
                raise L.OcrHipError("RCCL rendezvous: not every rank acknowledged the unique id within %d s (%s)"
                                    % (self.ACK_TIMEOUT_S, e)) from err
            if err is not None:
                raise err
            if bad:
                raise L.OcrHipError("RCCL rendezvous: rank(s) %s could not obtain the unique id" % bad)
        else:
            self._check(L._fn("ocr_comm_unique_id", ctypes.c_int)(ident), "ocr_comm_unique_id")
        self.handle = ctypes.c_void_p()
        self._check(L._fn("ocr_comm_init_rank", ctypes.c_int)(ctypes.byref(self.handle), ctypes.c_int(world),
                                                               ident, ctypes.c_int(rank)), "ocr_comm_init_rank")
        self.rank, self.world = rank, world

    @staticmethod
    def _check(rc, what):
        if rc != 0:
            from . import _lib as L
            raise L.OcrHipError("%s failed: %s (%s)" % (what, L.load().ocr_status_string(rc).decode(),
                                                        _last_comm_error()))

    def size(self):
        from . import _lib as L
        return L.call_int("ocr_comm_size", self.handle)

    def all_reduce_(self, t, op="sum", stream=None):
        """In-place all-reduce of a contiguous f32 / 16-bit / int32 device tensor on `stream` (default: current)."""
        import ctypes
        from . import _lib as L
        dt = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2, torch.int32: 3}[t.dtype]
        st = L.stream_ptr() if stream is None else ctypes.c_void_p(stream.cuda_stream)
        L.call("ocr_allreduce_bucket", self.handle, L.ptr(t), ctypes.c_size_t(t.numel()), ctypes.c_int(dt),
               ctypes.c_int({"sum": 0, "max": 1}[op]), st)

    def destroy(self):
        import ctypes
        from . import _lib as L
        if self.handle is not None and self.handle.value:
            L._fn("ocr_comm_destroy", ctypes.c_int)(self.handle)
            self.handle = None


def _last_comm_error():
    import ctypes
    from . import _lib as L
    fn = L._fn("ocr_comm_last_error", ctypes.c_char_p)
    return (fn() or b"").decode()


def exchange_mode(world, cuda, force=False):
    """"abi": buckets go through ocr_allreduce_bucket on this process's own RCCL communicator (every entry of
    the recorded step is then a C-ABI call); "torch": through torch.distributed (gloo on CPU / shared-GPU
    functional runs, or OCR_EXCHANGE=torch)."""
    import os
    want = os.environ.get("OCR_EXCHANGE", "")
    if want in ("abi", "torch"):
        return want
    if not cuda or (world == 1 and not force):
        return "torch"
    if td.is_available() and td.is_initialized() and td.get_backend() != "nccl":
        return "torch"
    return "abi"


class GradientAllReduce:
    def __init__(self, store, world_size, bucket_bytes=32 << 20, op="mean", group=None, fold_mean=False,
                 force=False, mode=None, comm=None, proxy=None, overlap=None):
        """op: "mean" = `average_gradients` (multigpu_train.py:70-85); "sum" = `sum_gradients`
        (train_pixellink.py:179-194: the caller has already divided its loss by num_clones).
        fold_mean: leave the SUM in the buffer and let the optimiser apply `grad_scale` (= 1/world)
        inside its own pass instead of one more sweep over the gradients.
        force: run the bucket / comm-stream / wait path at world 1 too (torch mode: a one-rank group must
        exist; abi mode: a one-rank communicator is created here).
        mode: "abi" | "torch" (default: `exchange_mode`).  In abi mode every bucket is
        event-record(compute) -> stream-wait(comm) -> ocr_allreduce_bucket(comm) -> event-record(comm), and
        `finish` is one stream-wait(compute) per bucket: C-ABI calls only, recorded into the step plan like
        any kernel launch.  In torch mode the hooks are host callbacks (`Recorder.py`).
        `enabled = False` turns every hook into a no-op (bench.py's comm-exposed A/B: the step without
        its exchange).
        proxy=(workgroups, link_gbps): abi mode only, a measurement aid — next to every bucket's all-reduce the step
        plan also holds an `ocr_comm_proxy` launch (include/ocr_hip.h: a stand-in with the shape of a multi-rank ring's
        device code) on the comm stream; a replayed step runs ONE of the two, chosen by `self.use_proxy` (bench.py's
        one-GPU `exchange.proxy` leg).
        overlap: True = a bucket's all-reduce is issued as soon as its last gradient is written, under the rest of
        backward; False = every bucket is issued after backward, in front of the optimiser (nothing shares the chip with
        the conv kernels; the whole exchange is exposed).  Default: OCR_EXCHANGE_OVERLAP (1).  Why the choice exists:
        DESIGN.md section 3.4 — the weight-gradient, persistent and 256-tile conv launches fill the chip in exactly one
        round of workgroups, so a comm kernel that holds even a few CUs sends their tiles into a second round.  With
        `proxy` (world 1, a measurement aid) the plan holds BOTH placements and a replay runs the one `self.overlap` names."""
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
        self.mode = (mode or exchange_mode(world_size, self.cuda, force)) if self.active else "torch"
        self.comm = comm
        self.proxy = proxy
        self.overlap = (__import__("os").environ.get("OCR_EXCHANGE_OVERLAP", "1") == "1") if overlap is None else bool(overlap)
        self.both_placements = proxy is not None and world_size == 1      # record early AND late launches (one-rank: harmless)
        assert not self.both_placements or world_size == 1     # (running a bucket's all-reduce twice is the identity only on one rank)
        self.use_proxy = False
        self.proxy_stats = None
        self.ev_ready = self.ev_done = None
        self._replaying = False
        if self.active and self.mode == "abi":
            import ctypes
            from . import _lib as L
            if self.comm is None:
                grouped = td.is_available() and td.is_initialized()
                rank = td.get_rank() if grouped else 0
                err = None
                # agree on availability BEFORE anybody enters the rendezvous: a rank that cannot load RCCL (or cannot
                # reach the key-value store) must not go to the collective below while the healthy ranks already sit in
                # ncclCommInitRank waiting for it — that would hang instead of falling back
                kv = None
                try:
                    if not L.call_int("ocr_comm_available"):
                        raise L.OcrHipError("RCCL is not available: %s" % _last_comm_error())
                    if grouped and world_size > 1:
                        kv = td.distributed_c10d._get_default_store()
                except Exception as e:
                    err = e
                if grouped and world_size > 1 and any_rank(err is not None):
                    err = err or L.OcrHipError("RCCL is not available on another rank")
                else:
                    try:
                        self.comm = AbiComm(rank, world_size, store=kv)
                    except Exception as e:                  # unique id / communicator refused, ...
                        err = e
                # every rank takes the SAME path: if any rank failed, all exchange through torch.distributed instead
                if grouped and world_size > 1 and any_rank(err is not None):
                    if self.comm is not None:
                        self.comm.destroy()
                        self.comm = None
                    import warnings
                    warnings.warn("C-ABI RCCL communicator unavailable (%r): gradient exchange through torch.distributed" % (err,))
                    self.mode = "torch"
                elif err is not None:
                    raise err
        if self.active and self.mode == "abi":
            import ctypes
            from . import _lib as L
            self.ev_ready, self.ev_done = [], []
            for _ in self.buckets:
                for lst in (self.ev_ready, self.ev_done):
                    h = ctypes.c_void_p()
                    L.check(L._fn("ocr_event_create", ctypes.c_int)(ctypes.byref(h)), "ocr_event_create")
                    lst.append(h)
        self.reset()

    def reset(self):
        self.left = list(self.need)
        self.handles = []
        self.fired = [False] * len(self.buckets)

    def bucket_nbytes(self):
        return [(e - s) * 4 for s, e in self.buckets]

    def on_grads_ready(self, variables):
        """Graph.backward's hook.  While a step is being recorded: in torch mode the hook itself becomes a
        host callback of the plan; in abi mode the C-ABI calls `_fire` makes are recorded (tagged "xchg")
        and replayed like any launch — the countdown below is record-time logic only."""
        if not self.active:
            return
        from . import _lib
        rec = _lib.RECORDER
        if rec is not None and self.mode == "torch" and not self._replaying:
            rec.py(lambda vs=tuple(variables): self._replayed(self.on_grads_ready, vs))
        if not self.enabled:
            return
        for v in variables:
            bi = self.bucket_of.get(v.name)
            if bi is None:
                continue
            self.left[bi] -= 1
            if self.left[bi] == 0 and not self.fired[bi] and (self.overlap or self.both_placements):
                self._fire(bi, "early")

    def _replayed(self, fn, *args):
        self._replaying = True
        try:
            fn(*args)
        finally:
            self._replaying = False

    def _fire(self, bi, when=None):
        """when: "early" / "late" = the bucket's placement (under backward / after it) when a plan records both
        (`both_placements`); None otherwise."""
        self.fired[bi] = True
        s, e = self.buckets[bi]
        buf = self.store.flat_grad[s:e]
        if self.mode == "abi":
            import ctypes
            from . import _lib as L
            cur = L.stream_ptr()
            cs = ctypes.c_void_p(self.comm_stream.cuda_stream)
            w = when if self.both_placements else None
            self._xcall("ocr_event_record", self.ev_ready[bi], cur, when=w)
            self._xcall("ocr_stream_wait_event", cs, self.ev_ready[bi], when=w)
            self._xcall("ocr_allreduce_bucket", self.comm.handle, L.ptr(buf), ctypes.c_size_t(e - s),
                        ctypes.c_int(0), ctypes.c_int(0), cs, kind="rccl", when=w)
            if self.proxy is not None:
                if self.proxy_stats is None:     # {min start, max end, tickets | busy ticks, launches}: ocr_comm_proxy
                    self.proxy_stats = torch.tensor([-1, 0, 0, 0, 0, 0, 0, 0], dtype=torch.int64).to(buf.device)
                self._xcall("ocr_comm_proxy", L.ptr(buf), ctypes.c_size_t((e - s) * 4), ctypes.c_int(int(self.proxy[0])),
                            ctypes.c_float(float(self.proxy[1])), L.ptr(self.proxy_stats), cs, kind="proxy", when=w)
            self._xcall("ocr_event_record", self.ev_done[bi], cs, when=w)
            if not any(h[0] == bi for h in self.handles):
                self.handles.append((bi, buf))
            return
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                h = td.all_reduce(buf, op=td.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            h = td.all_reduce(buf, op=td.ReduceOp.SUM, group=self.group, async_op=True)
        self.handles.append((h, buf))

    @staticmethod
    def _xcall(name, *args, kind=None, when=None):
        """A C-ABI call of the exchange, tagged ("xchg", kind, when) in a recorded plan: kind "rccl" / "proxy" = the two
        alternative comm-stream launches of a bucket, "finish" = the compute stream's wait for a bucket, None = event
        plumbing; when "early" / "late" = the placement the entry belongs to when the plan holds both (train.TrainStep.
        _replay runs the entries of the placement `overlap` names), None = always."""
        from . import _lib as L
        L.call(name, *args)
        if L.RECORDER is not None:
            L.RECORDER.tag_last(("xchg", kind, when))

    def finish(self):
        """Fire whatever has not been fired (variables without a gradient this step), wait for all
        buckets, and turn the sum into the tower mean (multigpu_train.py:80-81).  Recorded like
        `on_grads_ready`: a host callback in torch mode, C-ABI stream waits in abi mode."""
        from . import _lib
        rec = _lib.RECORDER
        if rec is not None and not self._replaying and (self.mode == "torch" or not self.active):
            rec.py(lambda: self._replayed(self.finish))
        if not self.active or not self.enabled:
            self.reset()
            return
        for bi in range(len(self.buckets)):
            if not self.fired[bi] or self.both_placements:
                self._fire(bi, "late")
        if self.mode == "abi":
            cur = _lib.stream_ptr()
            for bi, _ in self.handles:
                self._xcall("ocr_stream_wait_event", cur, self.ev_done[bi], kind="finish")
        else:
            for h, buf in self.handles:
                h.wait()          # cuda: makes the current stream wait for the collective
        if self.op == "mean" and not self.fold_mean:
            if self.cuda:
                from . import ops
                if self.mode == "abi":
                    # part of the exchange: replayed only while the exchange is enabled
                    self._xcall("ocr_scale_f32", _lib.ptr(self.store.flat_grad),
                                __import__("ctypes").c_int64(self.store.flat_grad.numel()),
                                __import__("ctypes").c_float(1.0 / self.world), _lib.stream_ptr())
                else:
                    # torch mode: the recorded host callback above re-runs finish() on every replay, so the scale
                    # itself must not ALSO enter the plan as a C entry (it would divide by the world size twice)
                    saved, _lib.RECORDER = _lib.RECORDER, None
                    try:
                        ops.scale_(self.store.flat_grad, 1.0 / self.world)
                    finally:
                        _lib.RECORDER = saved
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
