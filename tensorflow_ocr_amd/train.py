"""Training-step assembly: mirror of multigpu_train.py:27-36,70-85,103-142 (tower loss ->
mean of tower gradients -> Adam with staircase exponential decay -> EMA) and of
train_pixellink.py:179-194,222-243 (sum of pre-divided gradients -> Momentum).

One process drives one GPU ("tower"); the cross-tower gradient mean of `average_gradients`
(multigpu_train.py:70-85) is an RCCL all-reduce over the flat gradient buffer, issued by
`dist.GradientAllReduce` on a side stream.
"""
import math

import numpy as np
import torch

from . import ops
from .graph import F32

def exponential_decay(learning_rate, global_step, decay_steps=5000, decay_rate=0.94, staircase=True):
    """tf.train.exponential_decay (multigpu_train.py:104)."""
    e = global_step // decay_steps if staircase else global_step / decay_steps
    return learning_rate * decay_rate ** e


def pixellink_lr(global_step, base_lr=0.01):
    """train_pixellink.py:222-237: base_lr * {0.1 if step<20k, 0.01 if <40k, 0.001 if <60k, else 1}."""
    if global_step < 20000:
        f = 0.1
    elif global_step < 40000:
        f = 0.01
    elif global_step < 60000:
        f = 0.001
    else:
        f = 1.0
    return base_lr * f


def _regularization_loss(opt):
    """Sum of tf.GraphKeys.REGULARIZATION_LOSSES (multigpu_train.py:36): slim.l2_regularizer(wd) on
    every regularised variable = wd/2 * sum(w^2) over the head of the flat buffer.  One device
    reduction; returns a 1-element f32 device tensor (read it with .item() where the loss is logged)."""
    st = opt.g.store
    if getattr(opt, "_reg_out", None) is None:
        opt._reg_out = torch.zeros(1, dtype=F32, device=st.flat.device)
    if st.n_reg > 0 and opt.wd:
        opt.g.workspace()
        ops.sum_squares(st.flat[:st.n_reg], 0.5 * opt.wd, opt._reg_out, opt.g.ws_small)
    return opt._reg_out


class AdamOptimizer:
    """tf.train.AdamOptimizer + ExponentialMovingAverage(decay, num_updates=global_step) + slim L2
    regulariser gradient, as ONE fused launch over the tower's flat parameter buffer."""

    def __init__(self, graph, learning_rate=1e-4, decay_steps=5000, decay_rate=0.94,
                 moving_average_decay=0.997, weight_decay=1e-5, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.g = graph
        graph.ensure_materialised()
        st = graph.store
        self.m = torch.zeros_like(st.flat)
        self.v = torch.zeros_like(st.flat)
        self.ema = st.flat.clone() if moving_average_decay else None
        self.lr0, self.decay_steps, self.decay_rate = learning_rate, decay_steps, decay_rate
        self.mad, self.wd = moving_average_decay, weight_decay
        self.b1, self.b2, self.eps = beta1, beta2, epsilon
        self.global_step = 0

    def learning_rate(self):
        return exponential_decay(self.lr0, self.global_step, self.decay_steps, self.decay_rate, True)

    def apply_gradients(self, grad_scale=1.0):
        st = self.g.store
        t = self.global_step + 1
        lr = self.learning_rate()
        lr_t = lr * math.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)
        ema_d = min(self.mad, (1.0 + self.global_step) / (10.0 + self.global_step)) if self.mad else 0.0
        ops.adam_step(st.flat, st.flat_grad, self.m, self.v, self.ema, st.n_reg, lr_t, self.b1, self.b2,
                      self.eps, self.wd, grad_scale / self.g.loss_scale, ema_d)
        st.version += 1
        self.global_step += 1

    def shadow_state_dict(self):
        """EMA shadows by variable name (what test.py:149-150 restores)."""
        return self.g.store.by_variable(self.ema)

    # `tf.train.Saver(tf.global_variables())` (multigpu_train.py:144) also saves the optimiser's slot
    # variables `<var>/Adam`, `<var>/Adam_1`, the scalars `beta1_power`, `beta2_power` and
    # `global_step`; a resumed run continues the LR staircase, the bias correction and the EMA warm-up
    SLOTS = ("Adam", "Adam_1")

    def slot_state_dict(self):
        st = self.g.store
        return {"Adam": st.by_variable(self.m), "Adam_1": st.by_variable(self.v)}

    def scalar_state_dict(self):
        t = self.global_step
        return {"beta1_power": np.float32(self.b1 ** (t + 1)), "beta2_power": np.float32(self.b2 ** (t + 1))}

    def load_state(self, global_step=None, slots=None, ema=None):
        st = self.g.store
        if slots:
            st.load_by_variable(self.m, slots.get("Adam", {}))
            st.load_by_variable(self.v, slots.get("Adam_1", {}))
        if ema and self.ema is not None:
            st.load_by_variable(self.ema, ema)
        if global_step is not None:
            self.global_step = int(global_step)

    def regularization_loss(self):
        return _regularization_loss(self)


class MomentumOptimizer:
    """tf.train.MomentumOptimizer(lr, 0.9) with the PixelLink schedule (train_pixellink.py:222-243)."""

    def __init__(self, graph, base_lr=0.01, momentum=0.9, weight_decay=5e-4, moving_average_decay=None):
        self.g = graph
        graph.ensure_materialised()
        st = graph.store
        self.acc = torch.zeros_like(st.flat)
        self.ema = st.flat.clone() if moving_average_decay else None
        self.base_lr, self.momentum, self.wd, self.mad = base_lr, momentum, weight_decay, moving_average_decay
        self.global_step = 0

    def learning_rate(self):
        return pixellink_lr(self.global_step, self.base_lr)

    SLOTS = ("Momentum",)

    def shadow_state_dict(self):
        return self.g.store.by_variable(self.ema)

    def slot_state_dict(self):
        return {"Momentum": self.g.store.by_variable(self.acc)}

    def scalar_state_dict(self):
        return {}

    def load_state(self, global_step=None, slots=None, ema=None):
        st = self.g.store
        if slots:
            st.load_by_variable(self.acc, slots.get("Momentum", {}))
        if ema and self.ema is not None:
            st.load_by_variable(self.ema, ema)
        if global_step is not None:
            self.global_step = int(global_step)

    def regularization_loss(self):
        return _regularization_loss(self)

    def apply_gradients(self, grad_scale=1.0):
        st = self.g.store
        ema_d = min(self.mad, (1.0 + self.global_step) / (10.0 + self.global_step)) if self.mad else 0.0
        ops.momentum_step(st.flat, st.flat_grad, self.acc, self.ema, st.n_reg, self.learning_rate(),
                          self.momentum, self.wd, grad_scale / self.g.loss_scale, ema_d)
        st.version += 1
        self.global_step += 1


_VIEW_OPS = frozenset((
    "empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "view", "_unsafe_view", "reshape",
    "_reshape_alias", "slice", "select", "permute", "transpose", "t", "as_strided", "detach", "alias", "unsqueeze",
    "squeeze", "expand", "narrow", "unbind", "split", "split_with_sizes", "unfold", "view_as_real", "lift_fresh",
    "is_pinned", "_has_compatible_shallow_copy_type", "size", "stride", "sym_size", "sym_stride", "numel", "dim"))


def _record_audit():
    """Dispatch mode for the RECORDED step: the plan holds C-ABI calls only, so a torch operator that
    launches device work inside `forward_loss` (arithmetic on an input, a dtype cast, a `zeros`) would
    run once, at record time, and every replayed step would read its stale result.  Collects the names
    of such operators so `TrainStep` can refuse the recording; views and allocations are fine."""
    from torch.utils._python_dispatch import TorchDispatchMode

    class Audit(TorchDispatchMode):
        def __init__(self):
            super().__init__()
            self.offenders = []

        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            name = func.overloadpacket.__name__ if hasattr(func, "overloadpacket") else str(func)
            if name not in _VIEW_OPS:
                flat = list(args) + list((kwargs or {}).values()) + (list(out) if isinstance(out, (tuple, list)) else [out])
                if any(isinstance(t, torch.Tensor) and t.is_cuda for t in flat):
                    self.offenders.append(name)
            return out
    return Audit()


USE_GUESTS = __import__("os").environ.get("OCR_GUEST_STREAM", "1") == "1"


GUEST_COVER = float(__import__("os").environ.get("OCR_GUEST_COVER", "2.1"))
GUEST_PAIRED_GRID = int(__import__("os").environ.get("OCR_GUEST_PAIRED_GRID", "256"))
GUEST_MIN_US = float(__import__("os").environ.get("OCR_GUEST_MIN_US", "40"))
# a completed bucket's exchange entries at the NEXT fork instead of right behind the join that followed its last weight
# gradient: the comm kernel then starts beside held-back weight gradients (456 of 512 registers: the
# only kernels of the step it can share a CU with) instead of beside the next input-gradient launch (512: it cannot).
# Measured with the one-GPU stand-ins (bench.py exchange.proxy_at_fork): 1.34 -> 1.23 ms of the stand-in's 1.10-1.19 ms
# exposed per step, either footprint; on.  (Issuing an exchange entry later than recorded is always correct.)
XCHG_AT_FORK = __import__("os").environ.get("OCR_XCHG_AT_FORK", "1") == "1"
# hosts shared out with the guests still to come in view (round 6): the greedy choice covered conv2_1's apply pass with two
# weight gradients (791 us for a 577 us pass) and left conv1_2's pooled pass — the last and largest — 306 us of host for 585,
# 0.28 ms exposed on the critical path (profiles/r05_guest_pairs_trace.txt).  0: the round-5 greedy choice (A/B).
GUEST_BALANCE = __import__("os").environ.get("OCR_GUEST_BALANCE", "1") == "1"


def schedule_guests(entries, cover=None, min_us=None, xchg_at_fork=None, balance=None):
    """Run the HBM-bound batch-norm backward passes of the recorded step as GUESTS beside its weight gradients
    (csrc/guest_bn.hip; measured rates: profiles/r05_guest_pairs.json).

    Recorded order (layers._conv_backward, layers.conv2d): ... coefficients(L) ["pre"], apply(L) ["guest", bytes],
    dgrad(L), wgrad(L) = slab kernel ["side", FLOP] + slab sum ["reduce"], [exchange entries of the bucket that wgrad(L)
    completed], coefficients(L-1), apply(L-1), ...  The chain apply -> dgrad -> apply is the critical path; a weight
    gradient depends on its layer's apply pass only, and nothing but the optimiser (and the exchange) waits for it.
    So weight gradients are HELD BACK and spent as hosts: at every guest the plan forks, the guest goes to the second
    stream, and held-back slab kernels go to the main stream — chosen, oldest first, so that their estimated time
    covers `cover` x the guest's stand-alone time without overshooting it by much (beside a host a guest moves ~0.42 of
    its stand-alone rate, the host keeps ~0.9 of its own: small guests take small hosts, the deep layers' weight
    gradients — little HBM traffic of their own — are left for the large passes of conv2 / conv1, which come last) —
    then the plan joins: the next input-gradient kernel owns the whole register file and would only time-slice with a
    straggler.  The slab sums (and whatever was recorded behind them: a re-layout of the gradient, the exchange entries
    of the bucket it completed) follow the join: alone they take 13 us, beside a streaming guest 60-370.  A guest's
    waves (<= 56 registers per lane) are placed beside the weight gradient's resident workgroups (456 of 512).  What is
    still held back when something needs the gradients (the exchange's closing entries, the optimiser, any host
    callback) is issued there, in order; a guest with nothing to run beside stays on the main stream.  Exchange entries
    issued later than recorded are always correct: they order the comm stream behind the compute stream at that
    point."""
    cover = GUEST_COVER if cover is None else cover
    min_us = GUEST_MIN_US if min_us is None else min_us
    xchg_at_fork = XCHG_AT_FORK if xchg_at_fork is None else xchg_at_fork
    balance = GUEST_BALANCE if balance is None else balance
    out, pending, i, n = [], [], 0, len(entries)
    deferred = []                # exchange entries of hosts already run, waiting for the next fork (xchg_at_fork)

    def tag(e):
        return e[4] if (e[0] == "c" and e[4] is not None) else (None,)

    def travels(e):            # an exchange entry that belongs to the weight gradient in front of it
        t = tag(e)
        return t[0] == "xchg" and (len(t) < 2 or t[1] != "finish") and (len(t) < 3 or t[2] in (None, "early"))

    def guest_ahead(i0):             # is there a guest behind entry i0, before anything that needs the gradients?
        for q in range(i0 + 1, n):
            eq = entries[q]
            tq = tag(eq)
            if tq[0] == "guest":
                return True
            if eq[0] != "c" or (tq[0] == "xchg" and not travels(eq)):
                return False
        return False

    def outlook(i0):
        """(what the guests from entry i0 on want of hosts, us; what the weight gradients recorded behind i0 and in front of
        the last of those guests will supply, us)"""
        last = None
        for q in range(i0, n):
            tq = tag(entries[q])
            if tq[0] == "guest" and len(tq) > 1 and tq[1] / 5.0e6 >= min_us:
                last = q
        want = supply = 0.0
        if last is not None:
            for q in range(i0, last + 1):
                tq = tag(entries[q])
                if tq[0] == "guest" and len(tq) > 1 and tq[1] / 5.0e6 >= min_us:
                    want += cover * tq[1] / 5.0e6
                elif tq[0] == "side" and len(tq) >= 2 and q > i0:
                    supply += tq[1] / 1.3e9
        return want, supply

    def flush():
        out.extend(deferred)         # (recorded before anything still pending: their buckets are complete)
        del deferred[:]
        for head, tail, _ in pending:
            out.append(head)
            out.extend(tail)
        del pending[:]
    while i < n:
        e = entries[i]
        t = tag(e)
        if t[0] == "side" and len(t) < 2:
            out.append(e)                    # a weight gradient that cannot host (ops.conv2d_wgrad): stays in place
            i += 1
            continue
        if t[0] == "side" and not guest_ahead(i):
            out.extend(deferred)             # (no fork will come for them either)
            del deferred[:]
            out.append(e)                    # no guest left to host (a net without batch norm: every call stays in place,
            i += 1                           # and the exchange sees its buckets as early as it was recorded)
            continue
        if t[0] == "side":
            us = t[1] / 1.3e9                                     # FLOP at 1.3 PFLOP/s, in us
            j = i + 1
            while j < n and (tag(entries[j])[0] == "reduce" or (tag(entries[j])[0] == "side" and len(tag(entries[j])) < 2)
                             or travels(entries[j])):
                j += 1
            pending.append((e, entries[i + 1:j], us))
            i = j
            continue
        if t[0] == "guest":
            alone = (t[1] / 5.0e6) if len(t) > 1 else 0.0                  # bytes at 5 TB/s, in us
            if pending and alone >= min_us:          # (a fork + join costs the main queue ~25 us of idle time)
                need = cover * alone
                # a host that carries exchange entries (its weight gradient completed a bucket) may only go once every
                # weight gradient recorded before it has gone: the bucket's all-reduce reads all of them
                def closes_bucket(k):
                    return any(tag(x)[0] == "xchg" for x in pending[k][1])
                take, acc, skipped = [], 0.0, False
                if balance and len(pending) <= 12:
                    # what is exposed of a guest is what its hosts do not cover, and every host a guest can use the later
                    # guests can use too: when the guests still to come want more than all hosts can give, each gets its
                    # SHARE (target), and the subset of the pending hosts whose sum is closest to the target goes — a miss
                    # either way costs a guest (this one or a later one) the same; with hosts to spare, covering counts
                    want, supply = outlook(i)
                    have = sum(us for _, _, us in pending) + supply
                    scarce = have < want
                    target = need * have / want if scarce else need
                    best = None
                    for m in range(1, 1 << len(pending)):
                        ks = [k for k in range(len(pending)) if m >> k & 1]
                        if any(closes_bucket(k) and any(j not in ks for j in range(k)) for k in ks):
                            continue
                        tot = sum(pending[k][2] for k in ks)
                        miss = abs(tot - target) if scarce else (2.0 * (target - tot) if tot < target else 0.25 * (tot - target))
                        key = (miss, len(ks), ks)
                        if best is None or key < best[0]:
                            best = (key, ks, tot)
                    if best is not None and best[2] <= 3.0 * need:
                        take, acc = best[1], best[2]
                else:
                    for k, (_, _, us) in enumerate(pending):                  # oldest first, no large overshoot
                        if acc >= 0.85 * need:
                            break
                        if acc + us <= 1.3 * need and not (skipped and closes_bucket(k)):
                            take.append(k)
                            acc += us
                        else:
                            skipped = True
                if not take:
                    ok = [q for q in range(len(pending)) if q == 0 or not closes_bucket(q)]
                    k = min(ok, key=lambda q: pending[q][2])
                    if pending[k][2] <= 3.0 * need:
                        take = [k]
                if take:
                    g = list(e)
                    g[4] = ("guest", t[1] if len(t) > 1 else 0.0, "paired")
                    # the guest entry points' `max_workgroups` (second to last argument): one workgroup per CU beside hosts
                    if len(e[2]) >= 2 and t[-1] != "as_is":
                        g[2] = tuple(e[2][:-2]) + (__import__("ctypes").c_int(GUEST_PAIRED_GRID),) + (e[2][-1],)
                    out.append(["fork"])
                    out.extend(deferred)          # the exchange of buckets completed by EARLIER hosts starts with these hosts
                    del deferred[:]
                    out.append(g)
                    hosts = [pending[k] for k in take]
                    for k in reversed(take):
                        del pending[k]
                    for head, _, _ in hosts:
                        out.append(head)
                    out.append(["join"])
                    for _, tail, _ in hosts:
                        for te in tail:
                            (deferred if (xchg_at_fork and tag(te)[0] == "xchg") else out).append(te)
                    i += 1
                    continue
            out.append(e)
            i += 1
            continue
        if e[0] != "c" or t[0] == "xchg":
            # host callbacks (optimiser, torch-mode exchange), the exchange's closing entries — and an exchange entry that
            # TRAVELS behind a weight gradient which stayed in place (one that cannot host, or conv1_1's sums form): its
            # bucket's all-reduce reads every weight gradient recorded before it, the held-back ones included
            flush()
        out.append(e)
        i += 1
    flush()
    return out


class TrainStep:
    """One data-parallel training step of `multigpu_train.py`'s hot loop (:118-142,171-174) for
    this process's tower: forward -> loss -> backward (gradient buckets all-reduced while the rest
    of backward runs) -> optimiser + EMA.

    `forward_loss(graph, *batch) -> loss`.  The step has static shapes, buffers and launch order,
    so after two eager steps (variable creation, optimiser creation) the third is RECORDED as a
    flat list of C-ABI calls (`_lib.Recorder`) and every later step REPLAYS that list: the Python
    graph/tape logic, ~700 ctypes marshalling round trips and all per-step allocations disappear
    from the hot loop (the host was the bottleneck at ~40 ms/step).  New input data is copied into
    the recorded input buffers; `replay=False` keeps every step eager.

    Everything between the batch tensors and the loss must therefore be C-ABI calls: input
    preprocessing written with torch operators belongs in the input pipeline, before the step (where
    the reference has it too — its queues deliver preprocessed images).  The recording runs under an
    audit that raises if a torch operator touched device memory inside `forward_loss`."""

    def __init__(self, graph, forward_loss, optimizer_factory, world_size=1, bucket_bytes=32 << 20,
                 replay=True, grad_op="mean", force_reduce=False, comm_proxy=None):
        """grad_op: "mean" = multigpu_train.py's `average_gradients` (each tower differentiates its own
        loss, the gradients are averaged); "sum" = train_pixellink.py's `sum_gradients`: each clone
        differentiates loss / num_clones (:264) and the gradients are summed (:179-194).
        force_reduce: run the bucketed exchange at world 1 too (needs a one-rank process group)."""
        if grad_op not in ("mean", "sum"):
            raise ValueError("grad_op must be 'mean' or 'sum'")
        self.g = graph
        self.grad_op = grad_op
        self.force_reduce = force_reduce
        self.comm_proxy = comm_proxy      # (workgroups, link GB/s): dist.GradientAllReduce(proxy=...), a measurement aid
        self.backward_end_event = None    # bench.py: recorded on the compute stream in front of the first wait for a bucket
        if grad_op == "sum":
            graph.loss_div = float(world_size)       # total_clone_loss = sum(losses) / num_clones
        self.forward_loss = forward_loss
        self.optimizer_factory = optimizer_factory
        self.world = world_size
        self.bucket_bytes = bucket_bytes
        self.opt = None
        self.reducer = None
        self.replay = replay
        self.steps = 0
        self.plan = None
        self.static_batch = None
        self.loss = None
        self.guest_stream = self.guest_ptr = self.guest_event = None      # schedule_guests: the paired guest passes' stream
        self._packed_version = None

    def build(self, *batch):
        """Create the variables, the flat buffers, the optimiser and the reducer WITHOUT taking a
        step: one forward pass on `batch` whose side effects (BN moving-statistics update) are undone.
        After it `graph.store` / `self.opt` can be restored from a checkpoint before the first update
        (the reference restores before its first `sess.run(train_op)`, multigpu_train.py:153-158)."""
        if self.opt is not None:
            return self
        from .dist import GradientAllReduce
        g = self.g
        g.reset_tape()
        # the dry-run forward updates the BN moving statistics: put back what was there BEFORE it (variables that
        # exist already may hold loaded statistics; the ones the dry run creates start from their initial value)
        before = {n: g.store.vars[n].data.clone() for n in g.store.order if not g.store.vars[n].trainable}
        self.forward_loss(g, *batch)
        g.reset_tape()
        self.opt = self.optimizer_factory(g)
        g.store.reset_non_trainable()
        for n, t in before.items():
            g.store.vars[n].data.copy_(t)
        self.reducer = GradientAllReduce(g.store, self.world, self.bucket_bytes, op=self.grad_op,
                                         fold_mean=True, force=self.force_reduce, proxy=self.comm_proxy)
        return self

    # -- eager / recording path --------------------------------------------------------------
    def _eager(self, batch, record):
        from . import _lib
        g = self.g
        g.reset_tape()
        if record:
            rec = _lib.Recorder()
            _lib.RECORDER = rec
            g.keepalive = []
        audit = _record_audit() if record else __import__("contextlib").nullcontext()
        try:
            with audit:
                loss = self.forward_loss(g, *batch)
            if record and audit.offenders:
                raise RuntimeError(
                    "torch operators inside the recorded step would not be replayed: %s — move them "
                    "before the step (input pipeline) or use replay=False" % sorted(set(audit.offenders)))
            if self.opt is None:
                from .dist import GradientAllReduce
                self.opt = self.optimizer_factory(g)           # materialises the flat buffers
                self.reducer = GradientAllReduce(g.store, self.world, self.bucket_bytes, op=self.grad_op,
                                                 fold_mean=True, force=self.force_reduce, proxy=self.comm_proxy)
            g.backward(self.reducer.on_grads_ready if self.reducer.active else None)
            self.reducer.finish()           # records itself: host callback (torch mode) or C-ABI stream waits (abi mode)
        finally:
            _lib.RECORDER = None
        self.opt.apply_gradients(self.reducer.grad_scale)
        self._repack()                      # the conv operand packs of the new weights, one launch
        if record:
            rec.py(lambda: self.opt.apply_gradients(self.reducer.grad_scale))
            rec.entries[-1].append("opt")
            rec.py(self._repack)
            self.recorded = rec.entries                       # (kept: reschedule() derives another plan from the same recording)
            self.plan = schedule_guests(rec.entries) if USE_GUESTS else rec.entries
            self.static_batch = list(batch)
            self.keepalive, g.keepalive = g.keepalive, None
            g.reset_tape()
        self.loss = loss
        return loss

    def reschedule(self, **kw):
        """Another guest / exchange placement of the SAME recorded step (schedule_guests keywords): bench.py's A/B arms."""
        if self.plan is not None and USE_GUESTS:
            self.plan = schedule_guests(self.recorded, **kw)

    def _repack(self):
        """Batched re-pack of the [tap][cout][cin] / [tap][cin][cout] operand copies from the f32 masters, stamped
        with the store version it saw: a replayed plan holds no lazy refresh of these packs, so `_replay` compares
        the stamp and re-packs first when the masters changed outside the step (load_state_dict, a restored
        checkpoint, an EMA swap)."""
        self.g.repack_all()
        self._packed_version = self.g.store.version

    def _replay(self, batch):
        import ctypes
        from . import _lib, ops
        if self.g.store.version != self._packed_version:
            self._repack()
        for dst, src in zip(self.static_batch, batch):
            if src is not dst:
                dst.copy_(src, non_blocking=True)
        timing = ops.KERNEL_TIMING
        main = torch.cuda.current_stream()
        bwd_marked = False
        for e in self.plan:
            if e[0] == "fork":
                # a weight gradient and the guest pass paired with it (schedule_guests): the guest stream starts here
                if self.guest_stream is None:
                    self.guest_stream = torch.cuda.Stream()
                    self.guest_ptr = ctypes.c_void_p(self.guest_stream.cuda_stream)
                    self.guest_event = torch.cuda.Event()
                self.guest_event.record(main)
                self.guest_stream.wait_event(self.guest_event)
                continue
            if e[0] == "join":
                main.wait_stream(self.guest_stream)
                continue
            if e[0] == "c":
                tag = e[4]
                if tag is not None and tag[0] == "guest" and tag[-1] == "paired":
                    rc = e[1](*(e[2][:-1] + (self.guest_ptr,)))
                    if rc != 0:
                        _lib.check(rc, e[3])
                    continue
                if tag is not None and tag[0] == "xchg":
                    # the gradient exchange as C-ABI calls (dist.GradientAllReduce, abi mode)
                    if self.reducer.enabled:
                        kind = tag[1] if len(tag) > 1 else None
                        when = tag[2] if len(tag) > 2 else None
                        # a bucket's comm-stream launch: the RCCL all-reduce OR its one-GPU stand-in (dist: proxy)
                        if (kind == "proxy" and not self.reducer.use_proxy) or (kind == "rccl" and self.reducer.use_proxy):
                            continue
                        # a plan that holds both placements of the exchange (under backward / after it) runs one
                        if when is not None and when != ("early" if self.reducer.overlap else "late"):
                            continue
                        if (kind == "finish" or when == "late") and self.backward_end_event is not None and not bwd_marked:
                            self.backward_end_event.record(main)      # end of backward on the compute stream
                            bwd_marked = True
                        rc = e[1](*e[2])
                        if rc != 0:
                            _lib.check(rc, e[3])
                    continue
            if e[0] == "c":
                tag = e[4]
                if timing is not None and tag is not None and len(tag) > 2 and tag[0] not in ("side", "guest", "xchg"):      # (kernel instantiation, FLOP, phase)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    rc = e[1](*e[2])
                    e1.record()
                    timing.append((tag[0], tag[1], tag[2] if len(tag) > 2 else "", e0, e1))
                else:
                    rc = e[1](*e[2])
                if rc != 0:
                    _lib.check(rc, e[3])
            else:
                if len(e) > 2 and e[2] == "opt" and self.backward_end_event is not None and not bwd_marked:
                    self.backward_end_event.record(main)          # (no exchange in this step: backward ends here)
                    bwd_marked = True
                e[1]()
        return self.loss

    def __call__(self, *batch):
        self.steps += 1
        if self.plan is not None:
            return self._replay(batch)
        # step 1 creates variables, step 2 runs with the flat buffers; step 3 is steady state
        return self._eager(batch, record=self.replay and self.steps >= 3)
