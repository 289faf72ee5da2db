"""Host-side plumbing that stands in for the TF-1.4 graph/session the reference drives
(`tf.variable_scope`, `tf.get_variable`, `opt.compute_gradients`): a variable store with
TF-slim variable names, activation handles, and a reverse-mode tape.

Nothing here computes: every FLOP happens in libocr_hip.so via `ops`.  torch is used for
device memory only.
"""
import contextlib
import math

import numpy as np
import torch

from . import _lib as L
from . import ops

# F16 = the library's 16-bit storage dtype (IEEE half, or bfloat16 under OCR_STORAGE=bf16)
F16, F32 = (torch.bfloat16 if L.STORAGE == "bf16" else torch.float16), torch.float32


class Variable:
    """A TF variable: f32 master copy in the store's flat buffer, optional gradient view."""

    def __init__(self, name, shape, init, trainable, regularized):
        self.name = name
        self.shape = tuple(int(s) for s in shape)
        self.init = init            # numpy f32 until the store is materialised
        self.trainable = trainable
        self.regularized = regularized
        self.data = None            # torch f32 view
        self.grad = None            # torch f32 view (trainable only)
        self.packed = {}            # kind -> (version, tensor)

    @property
    def size(self):
        return int(np.prod(self.shape)) if self.shape else 1


class VariableStore:
    """Name -> Variable, materialised as flat f32 device buffers:
    [regularised trainables | other trainables] for params / grads (one Adam launch, one
    all-reduce bucket list) and a second flat buffer for non-trainables (BN moving stats)."""

    def __init__(self, device):
        self.vars = {}
        self.order = []
        self.device = device
        self.version = 0            # bumped by every optimiser step -> f16 packs are stale
        self.flat = self.flat_grad = self.flat_aux = None
        self.n_reg = 0

    def get(self, name, shape, initializer, trainable=True, regularized=False):
        v = self.vars.get(name)
        if v is not None:
            if tuple(shape) != v.shape:
                raise ValueError("variable %s exists with shape %s, requested %s" % (name, v.shape, tuple(shape)))
            return v
        if self.flat is not None:
            raise RuntimeError("variable %s created after the store was materialised" % name)
        init = np.ascontiguousarray(np.asarray(initializer(tuple(shape)), dtype=np.float32).reshape(shape))
        # non-trainables (BN moving statistics: small vectors) keep their initial value so that a
        # variable-creating dry run can be undone (train.TrainStep.build)
        v = Variable(name, shape, None if trainable else init, trainable, regularized)
        v.data = torch.from_numpy(init).to(self.device)
        if trainable:
            v.grad = torch.zeros_like(v.data)
        self.vars[name] = v
        self.order.append(name)
        return v

    def materialise(self):
        """Gather every variable into the flat buffers (idempotent); views replace the
        per-variable tensors used while the graph was being built."""
        if self.flat is not None:
            return
        tr = [self.vars[n] for n in self.order if self.vars[n].trainable]
        reg = [v for v in tr if v.regularized]
        oth = [v for v in tr if not v.regularized]
        aux = [self.vars[n] for n in self.order if not self.vars[n].trainable]

        def pack(vs, pad=4):
            offs, o = [], 0
            for v in vs:
                offs.append(o)
                o += (v.size + pad - 1) // pad * pad   # keep every view 16-byte aligned
            return offs, o

        offs_r, nr = pack(reg)
        offs_o, no = pack(oth)
        offs_a, na = pack(aux)
        self.n_reg = nr
        self.flat = torch.zeros(max(nr + no, 1), dtype=F32, device=self.device)
        self.flat_grad = torch.zeros_like(self.flat)
        self.flat_aux = torch.zeros(max(na, 1), dtype=F32, device=self.device)
        for v, o in list(zip(reg, offs_r)) + [(v, nr + o) for v, o in zip(oth, offs_o)]:
            view = self.flat[o:o + v.size].view(v.shape)
            view.copy_(v.data)
            gview = self.flat_grad[o:o + v.size].view(v.shape)
            gview.copy_(v.grad)
            v.data, v.grad = view, gview
        for v, o in zip(aux, offs_a):
            view = self.flat_aux[o:o + v.size].view(v.shape)
            view.copy_(v.data)
            v.data = view

    def trainable(self):
        return [self.vars[n] for n in self.order if self.vars[n].trainable]

    def reset_non_trainable(self):
        """Moving statistics back to their initial values (0 / 1)."""
        for n in self.order:
            v = self.vars[n]
            if not v.trainable and v.init is not None:
                v.data.copy_(torch.from_numpy(v.init))

    def by_variable(self, flat_like):
        """{variable name: numpy copy} of a buffer laid out like `flat` (gradients, optimiser slots, EMA)."""
        base = self.flat.data_ptr()
        out = {}
        for v in self.trainable():
            off = (v.data.data_ptr() - base) // 4
            out[v.name] = flat_like[off:off + v.size].view(v.shape).detach().cpu().numpy().copy()
        return out

    def load_by_variable(self, flat_like, sd):
        """Inverse of `by_variable` for the names present in `sd`; returns the names loaded."""
        base = self.flat.data_ptr()
        done = []
        for v in self.trainable():
            if v.name in sd:
                off = (v.data.data_ptr() - base) // 4
                a = np.asarray(sd[v.name], dtype=np.float32).reshape(v.shape)
                flat_like[off:off + v.size].view(v.shape).copy_(torch.from_numpy(a))
                done.append(v.name)
        return done

    def state_dict(self):
        return {n: self.vars[n].data.detach().cpu().numpy().copy() for n in self.order}

    def load_state_dict(self, sd, strict=True):
        for n, arr in sd.items():
            if n not in self.vars:
                if strict:
                    raise KeyError(n)
                continue
            v = self.vars[n]
            a = np.asarray(arr, dtype=np.float32).reshape(v.shape)
            v.data.copy_(torch.from_numpy(a))
        self.version += 1


class Act:
    """Activation handle: device tensor + (lazily created) gradient.

    `data` may be DEFERRED: a ResNet bottleneck's output relu(bn(conv3) + shortcut) is not computed when the unit is
    built — the next 1x1 convolution that consumes it computes it while loading its operand and writes it back
    (resnet_layers.conv_bn_raw, ocr_conv2d_pw_bnaddrelu_f16).  Any other access to `.data` runs the plain
    element-wise pass first, so consumers that know nothing about this stay correct."""

    __slots__ = ("_data", "grad", "requires_grad", "name", "bn_ctx", "bn_partial", "tail_ctx", "tail_partial",
                 "pending", "sub_grad", "deferred", "tail_fwd", "pending_owner", "consumed", "bn_fwd", "bias_ctx",
                 "bias_partial", "sole_consumer", "bias_done", "pool_grad", "takes_pool_grad", "first_s1")

    def __init__(self, data, requires_grad=True, name=""):
        self._data = data
        self.grad = None
        self.requires_grad = requires_grad
        self.name = name
        self.bn_ctx = None        # (y, scale, shift, mean, invstd, relu) of the conv+BN that made it
        self.bn_partial = None    # (partial, T): BN-backward sums already reduced by the dgrad conv
        # ResNet bottleneck outputs (resnet_layers.bottleneck): (y, mean, invstd of the unit's last conv, own data, mask
        # bits); the next unit's LAST gradient contribution (counted down in `pending`, by the consumers that carry
        # `pending_owner`) then stores the gradient past this output's ReLU and leaves the BN-backward sums in
        # tail_partial = (partial, T)
        self.tail_ctx = None
        self.tail_partial = None
        self.pending = None
        self.pending_owner = None
        self.consumed = False     # a convolution has read it already (tail fusion must then stay off: build order)
        self.sub_grad = None      # gradient of this output's stride-2 subsample, waiting for the fused tail conv
        self.deferred = None      # callable that fills _data (the plain pass), while nobody has computed it yet
        self.bias_ctx = None      # bias + ReLU producers: bn_ctx-shaped tuple (activation, 1, 0, 0, 1, relu) for the consumer's epilogue
        self.bias_partial = None  # (partial, T): the consumer stored dz and left the bias-gradient partial sums
        self.bias_done = False    # the layer's own pool already produced dz and the bias gradient (layers.max_pool2d)
        self.sole_consumer = False  # set by the NET for activations read by exactly one convolution (nets/vgg.py)
        self.takes_pool_grad = False   # the producer's backward can gather its gradient from a max-pool's operands ...
        self.pool_grad = None     # ... which that pool's backward then leaves here: (pooled grad, argmax, k, stride, (pt, pl))
        self.first_s1 = None      # conv1_1's activation: the sums its weight gradient is finished from, left by the sole consumer's
                                  # input-gradient launch INSTEAD of the gradient (layers._conv_dgrad; `grad` then stays None)
        self.bn_fwd = None        # (y, scale, shift, relu): a deferred relu(bn(y)) a fusing consumer (max-pool) can evaluate itself
        self.tail_fwd = None      # (y3, scale, shift, shortcut tensor, sc_scale, sc_shift, bits): what a fusing consumer needs

    @property
    def data(self):
        if self.deferred is not None:
            fill, self.deferred = self.deferred, None
            fill()
        return self._data

    @data.setter
    def data(self, value):
        self._data = value

    @property
    def shape(self):
        return tuple(self._data.shape)


class Graph:
    """One tower: variable store + tape + scratch.  `loss_scale` multiplies the loss gradient so
    that f16 activation gradients stay in range; the optimiser divides it out."""

    def __init__(self, device="cuda:0", loss_scale=1024.0, seed=1, precision="f16"):
        """precision: "f16" = the product path (f16 storage, f32 accumulate, MFMA kernels);
        "f32" = forward-only f32 INFERENCE precision (f32 storage and arithmetic, the convolutions on the matrix
        cores with v_mfma_f32_32x32x2_f32: layers_f32.py, csrc/f32_infer.hip): whole-graph outputs within 1e-3 of the
        f32 reference (the north star's tolerance), ~10x the time of the 16-bit path."""
        if precision not in ("f16", "f32"):
            raise ValueError("precision must be 'f16' or 'f32'")
        self.precision = precision
        self.device = torch.device(device)
        self.store = VariableStore(self.device)
        self.tape = []
        self.scope = []
        self.loss_scale = float(loss_scale)
        self.loss_div = 1.0          # train_pixellink.py:264: each clone's loss is divided by num_clones
        self.rng = np.random.default_rng(seed)
        self.ws = None
        self.ws_small = None
        self.ws_wgrad = None
        self.collections = {"losses": [], "update_ops": []}
        self.keepalive = None        # list while a step is being recorded (train.TrainStep)

    # --- scopes / variables (tf.variable_scope, tf.get_variable) ---
    @contextlib.contextmanager
    def variable_scope(self, name):
        if name:
            self.scope.append(name)
        try:
            yield
        finally:
            if name:
                self.scope.pop()

    def full_name(self, name):
        return "/".join(self.scope + [name])

    def get_variable(self, name, shape, initializer, trainable=True, regularized=False):
        return self.store.get(self.full_name(name), shape, initializer, trainable, regularized)

    def seed_scale(self):
        """d(loss)/d(loss) the loss kernels start the backward pass from: the f16 loss scale, over
        the clone count when the caller pre-divides its loss (grad_op="sum")."""
        return self.loss_scale / self.loss_div

    # --- device memory ---
    def workspace(self):
        if self.ws is None:
            self.ws = ops.Workspace(self.device, 256 << 20)
            self.ws_small = ops.Workspace(self.device, 8 << 20)
            self.ws_wgrad = ops.Workspace(self.device, 160 << 20)   # weight-gradient slabs (side stream)
        return self.ws

    def empty(self, shape, dtype=F16):
        t = torch.empty(shape, dtype=dtype, device=self.device)
        if self.keepalive is not None:
            self.keepalive.append(t)
        return t

    def zeros(self, shape, dtype=F32):
        """Zero-filled scratch.  NOT replay-safe inside a recorded step (the fill is a torch op):
        ops that need zeros on every step must clear through a recorded kernel."""
        t = torch.zeros(shape, dtype=dtype, device=self.device)
        if self.keepalive is not None:
            self.keepalive.append(t)
        return t

    def ensure_materialised(self):
        self.store.materialise()

    def packed(self, var, kind, fn):
        """f16 re-pack of a variable, refreshed when the optimiser has stepped."""
        ent = var.packed.get(kind)
        if ent is None or ent[0] != self.store.version:
            t = fn(ent[1] if ent is not None else None)
            var.packed[kind] = (self.store.version, t)
            return t
        return ent[1]

    def repack_all(self):
        """Refresh every [tap][cout][cin] / [tap][cin][cout] operand pack — and the fuse heads' zero-padded [32][cin] /
        [cin][32] pairs — in ONE launch (after the optimiser step; the packs' own lazy refresh in `packed` then finds
        them current).  Other pack kinds stay lazy."""
        from . import ops
        ents = []
        for v in self.store.vars.values():
            pk = getattr(v, "packed", {})
            if "kc_ck" in pk:
                ents.append((v, "kc_ck", pk["kc_ck"], 0))
            if "small" in pk:
                ents.append((v, "small", pk["small"], 32))
        if not ents:
            return
        key = tuple(id(e[2][1]) for e in ents)
        pb = getattr(self, "_pack_batch", None)
        if pb is None or pb[0] != key:
            pb = (key, ops.PackBatch([(v.data, e[1][0], e[1][1], ld) for v, _, e, ld in ents], self.device))
            self._pack_batch = pb
        pb[1].run()
        for v, kind, e, _ in ents:
            v.packed[kind] = (self.store.version, e[1])

    # --- tape ---
    def record(self, fn, produces=()):
        """`produces`: the variables whose gradients are complete once `fn` has run."""
        self.tape.append((fn, tuple(produces)))

    def backward(self, on_grads_ready=None):
        """Run the tape in reverse (the loss op seeds its own gradient).  `on_grads_ready(vars)` is
        called as soon as a closure has finished a set of parameter gradients — the hook the
        data-parallel all-reduce uses to overlap communication with the rest of backward."""
        if self.precision != "f16":
            raise NotImplementedError("the f32 inference precision is forward-only")
        for fn, produces in reversed(self.tape):
            fn()
            if on_grads_ready is not None and produces:
                on_grads_ready(produces)      # while a step is recorded the hook records itself (dist.py)
        self.tape.clear()

    def reset_tape(self):
        self.tape.clear()
        self.collections["losses"].clear()


_default = None


def get_default_graph():
    global _default
    if _default is None:
        _default = Graph()
    return _default


def set_default_graph(g):
    global _default
    _default = g
    return g


# --- initialisers (slim defaults) -------------------------------------------------------
def xavier_uniform(rng):
    def init(shape):
        if len(shape) == 4:
            fan_in, fan_out = shape[0] * shape[1] * shape[2], shape[0] * shape[1] * shape[3]
        else:
            fan_in, fan_out = shape[0], shape[-1]
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return rng.uniform(-lim, lim, size=shape)
    return init


def variance_scaling(rng, factor=2.0):
    """slim.variance_scaling_initializer(): truncated normal, FAN_IN, factor 2."""
    def init(shape):
        fan_in = shape[0] * shape[1] * shape[2] if len(shape) == 4 else shape[0]
        std = math.sqrt(1.3 * factor / fan_in)
        x = rng.normal(0.0, std, size=shape)
        bad = np.abs(x) > 2 * std
        while bad.any():
            x[bad] = rng.normal(0.0, std, size=int(bad.sum()))
            bad = np.abs(x) > 2 * std
        return x
    return init


def constant(value):
    return lambda shape: np.full(shape, value, dtype=np.float32)
