"""Thin, typed wrappers over the C ABI (include/ocr_hip.h).

Every function takes torch CUDA tensors purely as device-memory handles, passes raw
pointers + the current HIP stream through ctypes, and returns nothing new except
tensors it was asked to allocate.  There is no torch compute and no CPU fallback
here: without libocr_hip.so every call raises `_lib.OcrHipError`.
"""
import ctypes
from ctypes import byref, c_double, c_float, c_int, c_int64, c_size_t

import torch

from . import _lib as L
from ._lib import ConvDesc, CONV_ACCUM_F16, CONV_BIAS, CONV_RELU, CONV_STATS, ptr

F16, F32 = (torch.bfloat16 if L.STORAGE == "bf16" else torch.float16), torch.float32


def _st():
    return L.stream_ptr()


def same_pad(size, k, stride, dilation=1):
    """TF 'SAME' padding for one spatial dim -> (out, pad_before)."""
    k_eff = (k - 1) * dilation + 1
    out = -(-size // stride)
    total = max((out - 1) * stride + k_eff - size, 0)
    return out, total // 2


class Workspace:
    """Stream-ordered scratch arena shared by all kernels of one tower."""

    def __init__(self, device, nbytes=64 << 20):
        self.device = device
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.retired = []

    def get(self, nbytes):
        if nbytes > self.buf.numel():
            # a recorded step plan or a captured HIP graph may hold raw pointers into the old arena:
            # it stays allocated (never handed back to the caching allocator) for the tower's lifetime
            self.retired.append(self.buf)
            self.buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=self.device)
        return self.buf

    def two(self, a_bytes, b_bytes):
        """Two disjoint 256-byte aligned regions."""
        a_al = (a_bytes + 255) // 256 * 256
        buf = self.get(a_al + b_bytes)
        return buf[:a_al], buf[a_al:a_al + b_bytes]


# ----------------------------------------------------------------------------- conv
def conv_desc(x_shape, cout, kh, kw, stride=1, dilation=1, pad=None, out_hw=None, flags=0,
              flip=0):
    n, h, w, cin = x_shape
    if pad is None:
        oh, pt = same_pad(h, kh, stride, dilation)
        ow, pl = same_pad(w, kw, stride, dilation)
    else:
        pt, pl = pad
        oh, ow = out_hw
    return ConvDesc(n, h, w, cin, oh, ow, cout, kh, kw, stride, dilation, pt, pl, flip, flags)


def conv2d_num_mtiles(d):
    return L.call_int("ocr_conv2d_num_mtiles", byref(d))


# When bench.py sets KERNEL_TIMING to a list, every implicit-GEMM conv launch is bracketed by
# HIP events on the launch stream and logged as (variant, flops, start_event, stop_event).
KERNEL_TIMING = None


_VARIANTS = {}


def conv2d_variant(d):
    """Which conv_igemm_kernel<BN,CK,WCO,M16,TH> instantiation the library launches for `d`."""
    key = (d.n, d.h, d.w, d.cin, d.cout, d.kh, d.kw, d.stride, d.dilation)
    v = _VARIANTS.get(key)
    if v is None:
        import ctypes
        buf = ctypes.create_string_buffer(96)
        L.call_int("ocr_conv2d_variant", byref(d), buf, ctypes.c_size_t(96))
        v = _VARIANTS[key] = buf.value.decode()
    return v


def conv2d(d, x, w_kc, y, bias=None, stats=None):
    flops = 2.0 * d.n * d.oh * d.ow * d.cout * d.cin * d.kh * d.kw
    # tag: (kernel instantiation, FLOP, phase) -- phase "fwd" launches run with nothing beside them,
    # "dgrad" launches share the GPU with the side-stream weight gradients
    tag = (conv2d_variant(d), flops, "dgrad" if d.flip_taps else "fwd")
    if KERNEL_TIMING is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.call("ocr_conv2d_f16", byref(d), ptr(x), ptr(w_kc), ptr(bias), ptr(y), ptr(stats), _st())
        e1.record()
        KERNEL_TIMING.append(tag + (e0, e1))
    else:
        L.call("ocr_conv2d_f16", byref(d), ptr(x), ptr(w_kc), ptr(bias), ptr(y), ptr(stats), _st())
    if L.RECORDER is not None:
        L.RECORDER.tag_last(tag)


def conv2d_relu_pool(d, x, w_kc, bias, pooled, argmax=None):
    """conv3x3 + bias + ReLU + 2x2/2 max-pool in one kernel (64 -> 64 channels): pooled output (+ first-max positions) only."""
    flops = 2.0 * d.n * d.oh * d.ow * d.cout * d.cin * d.kh * d.kw
    tag = (conv2d_variant(d), flops, "fwd")
    if KERNEL_TIMING is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.call("ocr_conv2d_relu_pool_f16", byref(d), ptr(x), ptr(w_kc), ptr(bias), ptr(pooled), ptr(argmax), _st())
        e1.record()
        KERNEL_TIMING.append(tag + (e0, e1))
    else:
        L.call("ocr_conv2d_relu_pool_f16", byref(d), ptr(x), ptr(w_kc), ptr(bias), ptr(pooled), ptr(argmax), _st())
    if L.RECORDER is not None:
        L.RECORDER.tag_last(tag)


def conv2d_bnred(d, x, w_kc, y, partial, bn_ctx, store_masked=False):
    """Input-gradient conv fused with the BN-backward reduction of the layer below.  store_masked: y receives the
    gradient PAST that layer's ReLU (bias nets: bn_ctx = (activation, ones, zeros, zeros, ones, True))."""
    by, sc, sh, mu, istd, relu = bn_ctx
    if isinstance(by, LazyFirstY):
        by = by.tensor()                 # a consumer that cannot recompute conv1_1's y: evaluate and store it now
    L.call("ocr_conv2d_bnred_f16", byref(d), ptr(x), ptr(w_kc), ptr(y), ptr(partial), ptr(by), ptr(sc),
           ptr(sh), ptr(mu), ptr(istd), c_int(int(relu)), c_int(int(store_masked)), _st())
    if L.RECORDER is not None:
        flops = 2.0 * d.n * d.oh * d.ow * d.cout * d.cin * d.kh * d.kw
        L.RECORDER.tag_last((conv2d_variant(d), flops, "dgrad"))


class LazyFirstY:
    """conv1_1's raw output y, which the batch-norm training step no longer stores (the statistics pass writes nothing,
    the activation and both readers of y in the backward pass evaluate the 3-channel convolution again): `tensor()`
    evaluates and stores it for a consumer that cannot."""

    def __init__(self, alloc, x4, w_first, shape):
        self.alloc, self.x4, self.w_first, self.shape = alloc, x4, w_first, shape
        self.t = None
        self.moments = None       # f64 [32*32]: the image's patch moments, when the statistics came from them (kept for the backward)

    def tensor(self):
        if self.t is None:
            self.t = self.alloc(self.shape)
            conv2d_first(self.x4, self.w_first, self.t)
        return self.t


def conv2d_bnred_first(d, x, w_kc, y, partial, first_ctx):
    """conv2d_bnred for the consumer of conv1_1's activation with conv1_1's y recomputed from the image instead of read:
    first_ctx = (x4, w_first, scale, shift, mean, invstd, relu)."""
    x4, wf, sc, sh, mu, istd, relu = first_ctx
    L.call("ocr_conv2d_bnred_first_f16", byref(d), ptr(x), ptr(w_kc), ptr(y), ptr(partial), ptr(x4), ptr(wf), ptr(sc),
           ptr(sh), ptr(mu), ptr(istd), c_int(int(relu)), _st())
    if L.RECORDER is not None:
        flops = 2.0 * d.n * d.oh * d.ow * d.cout * d.cin * d.kh * d.kw
        L.RECORDER.tag_last((conv2d_variant(d), flops, "dgrad"))


def conv2d_bnred_first_wgrad_blocks(d):
    """Number of S1 blocks conv2d_bnred_first_wgrad leaves (one per workgroup); -1 where that launch is unsupported."""
    return int(L._fn("ocr_conv2d_bnred_first_wgrad_blocks", ctypes.c_int)(byref(d)))


def conv2d_bnred_first_wgrad(d, x, w_kc, partial, first_ctx, s1):
    """conv2d_bnred_first that does not store the gradient: `s1` [blocks][32][64] f32 receives the sums conv1_1's weight
    gradient is finished from (conv2d_first_wgrad_sums)."""
    x4, wf, sc, sh, mu, istd, relu = first_ctx
    L.call("ocr_conv2d_bnred_first_wgrad_f16", byref(d), ptr(x), ptr(w_kc), ptr(partial), ptr(x4), ptr(wf), ptr(sc),
           ptr(sh), ptr(mu), ptr(istd), c_int(int(relu)), ptr(s1), _st())
    if L.RECORDER is not None:
        flops = 2.0 * d.n * d.oh * d.ow * d.cout * d.cin * d.kh * d.kw
        L.RECORDER.tag_last((conv2d_variant(d), flops, "dgrad"))


def conv2d_first_wgrad_sums(s1, moments, w_first, coef, dw):
    """dW of conv1_1 = A .* S1 + B .* (M W) + C .* m (csrc/conv_first.hip: first_wgrad_sums_kernel)."""
    a, b, c = coef
    L.call("ocr_conv2d_first_wgrad_sums_f32", ptr(s1), c_int(s1.shape[0]), ptr(moments), ptr(w_first),
           c_int(dw.shape[-1]), ptr(a), ptr(b), ptr(c), ptr(dw), _st())


def conv2d_bnred_tail(d, x, w_kc, y, partial, tail_ctx, sub_grad=None):
    """Input-gradient conv that completes the gradient of a bottleneck output: stores the gradient past the
    output ReLU and emits the BN-backward sums of the unit's last conv; `sub_grad`: gradient of the output's
    stride-2 subsample, added at the even positions (include/ocr_hip.h).  tail_ctx = (y, mean, invstd, out[, bits]):
    with `bits` (the output's ReLU mask, one byte per 8 channels) the kernel reads them instead of `out`."""
    by, mu, istd, out = tail_ctx[:4]
    bits = tail_ctx[4] if len(tail_ctx) > 4 else None
    L.call("ocr_conv2d_bnred_tail_f16", byref(d), ptr(x), ptr(w_kc), ptr(y), ptr(partial), ptr(by), ptr(mu),
           ptr(istd), ptr(out), ptr(bits), ptr(sub_grad), _st())
    if L.RECORDER is not None:
        flops = 2.0 * d.n * d.oh * d.ow * d.cout * d.cin * d.kh * d.kw
        L.RECORDER.tag_last((conv2d_variant(d), flops, "dgrad"))


def conv2d_pw_bnaddrelu(d, prev_y, prev_scale, prev_shift, shortcut, sc_scale, sc_shift, x_out, bits, w_kc, y, stats):
    """1x1 conv whose input x = relu(bn(prev_y) + shortcut) is computed while it is loaded and written to x_out (+ bits)."""
    L.call("ocr_conv2d_pw_bnaddrelu_f16", byref(d), ptr(prev_y), ptr(prev_scale), ptr(prev_shift), ptr(shortcut),
           ptr(sc_scale), ptr(sc_shift), ptr(x_out), ptr(bits), ptr(w_kc), ptr(y), ptr(stats), _st())
    if L.RECORDER is not None:
        flops = 2.0 * d.n * d.oh * d.ow * d.cout * d.cin
        L.RECORDER.tag_last((conv2d_variant(d).replace("conv_pw_kernel", "conv_pwx_kernel"), flops, "fwd"))


def conv2d_pw_bnrelu(d, prev_y, prev_scale, prev_shift, x_out, w_kc, y, stats):
    """1x1 conv whose input x = relu(bn(prev_y)) is computed while it is loaded and written to x_out."""
    L.call("ocr_conv2d_pw_bnrelu_f16", byref(d), ptr(prev_y), ptr(prev_scale), ptr(prev_shift), ptr(x_out), ptr(w_kc), ptr(y),
           ptr(stats), _st())
    if L.RECORDER is not None:
        flops = 2.0 * d.n * d.oh * d.ow * d.cout * d.cin
        L.RECORDER.tag_last((conv2d_variant(d).replace("conv_pw_kernel", "conv_pwx_kernel"), flops, "fwd"))


def conv2d_pw_bnbwd_bnred(d, dz, y_above, coef, dy_out, w_kc, dx, partial, bn_ctx):
    """1x1 input-gradient conv whose operand dy = A*dz + B*y_above + C is computed while it is loaded and written to
    dy_out; fused BN-backward reduction of the layer below as in conv2d_bnred."""
    by, sc, sh, mu, istd, relu = bn_ctx
    if isinstance(by, LazyFirstY):
        by = by.tensor()
    a, b, c = coef
    L.call("ocr_conv2d_pw_bnbwd_bnred_f16", byref(d), ptr(dz), ptr(y_above), ptr(a), ptr(b), ptr(c), ptr(dy_out),
           ptr(w_kc), ptr(dx), ptr(partial), ptr(by), ptr(sc), ptr(sh), ptr(mu), ptr(istd), c_int(int(relu)), _st())
    if L.RECORDER is not None:
        flops = 2.0 * d.n * d.oh * d.ow * d.cout * d.cin
        L.RECORDER.tag_last((conv2d_variant(d).replace("conv_pw_kernel", "conv_pwx_kernel"), flops, "dgrad"))


def conv2d_pw_bnbwd_tail(d, dz, y_above, coef, relu_shift, dy_out, w_kc, dx, partial=None, tail_ctx=None, sub_grad=None):
    """conv2d_pw_bnbwd_bnred's loader (+ the ReLU mask of the BN above when relu_shift is given) in front of the plain /
    accumulating (d.flags) / bottleneck-tail epilogue (tail_ctx, sub_grad as conv2d_bnred_tail)."""
    a, b, c = coef
    by = mu = istd = out = bits = None
    if tail_ctx is not None:
        by, mu, istd, out = tail_ctx[:4]
        bits = tail_ctx[4] if len(tail_ctx) > 4 else None
    L.call("ocr_conv2d_pw_bnbwd_tail_f16", byref(d), ptr(dz), ptr(y_above), ptr(a), ptr(b), ptr(c), ptr(relu_shift),
           ptr(dy_out), ptr(w_kc), ptr(dx), ptr(partial), ptr(by), ptr(mu), ptr(istd), ptr(out), ptr(bits), ptr(sub_grad),
           _st())
    if L.RECORDER is not None:
        flops = 2.0 * d.n * d.oh * d.ow * d.cout * d.cin
        L.RECORDER.tag_last((conv2d_variant(d).replace("conv_pw_kernel", "conv_pwx_kernel"), flops, "dgrad"))


def bn_bwd_sums(partial, T, c, out0, out1, ws):
    """Column sums of fused-reduction partial rows [T][2][c] (kind 0 -> out0, kind 1 -> out1)."""
    stage = ws.get(bn_reduce_workspace(T, c))
    L.call("ocr_bn_bwd_sums", ptr(partial), c_int(T), c_int(c), ptr(out0), ptr(out1), ptr(stage),
           c_size_t(stage.numel()), _st())


def bn_bwd_coefficients(partial, T, c, count, scale, save_mean, save_invstd, dgamma, dbeta, coef, ws):
    stage = ws.get(bn_reduce_workspace(T, c))
    a, b, cc = coef
    L.call("ocr_bn_bwd_coefficients", ptr(partial), c_int(T), c_int(c), c_double(count), ptr(scale), ptr(save_mean),
           ptr(save_invstd), ptr(dgamma), ptr(dbeta), ptr(a), ptr(b), ptr(cc), ptr(stage), c_size_t(stage.numel()), _st())


def bn_relu_bwd_apply(y, scale, shift, save_mean, save_invstd, da_full, relu, partial, T, dgamma, dbeta, dy, ws):
    n, h, w, c = y.shape
    stage = ws.get(bn_reduce_workspace(T, c))
    L.call("ocr_bn_relu_bwd_apply_f16", ptr(y), ptr(scale), ptr(shift), ptr(save_mean), ptr(save_invstd),
           ptr(da_full), c_int(n), c_int(h), c_int(w), c_int(c), c_int(int(relu)), ptr(partial), c_int(T),
           ptr(dgamma), ptr(dbeta), ptr(dy), ptr(stage), c_size_t(stage.numel()), _st())


# --- guest forms (csrc/guest_bn.hip): <= 56 registers per lane, one launch, placed beside a resident weight-gradient
# workgroup; the recorded step runs them on a second stream (train.TrainStep: tags "pre" / "guest") -------------------
GUEST_BN = __import__("os").environ.get("OCR_GUEST_BN", "1") == "1"


def guest_apply_ok(shape):
    """The guest kernels address through 32-bit buffer offsets: tensors < 2 GiB, power-of-two channel chunks."""
    n, h, w, c = shape
    q = c // 4
    return GUEST_BN and c % 8 == 0 and q <= 256 and (q & (q - 1)) == 0 and n * h * w * c * 2 < (1 << 31)


def bn_bwd_coefficients_pre(partial, T, c, count, scale, save_mean, save_invstd, dgamma, dbeta, coef, ws):
    """ocr_bn_bwd_coefficients in front of a guest apply pass: tagged so that the recorded step issues it BEFORE the
    weight gradient the guest runs beside (its finalize workgroups need 72 registers: queued behind a resident
    weight-gradient grid they would hold the guest back for that grid's whole launch)."""
    bn_bwd_coefficients(partial, T, c, count, scale, save_mean, save_invstd, dgamma, dbeta, coef, ws)
    if L.RECORDER is not None:
        L.RECORDER.tag_last(("pre",))


def bn_relu_bwd_apply_affine(y, da, scale, shift, coef_b, coef_c, relu, dy):
    n, h, w, c = y.shape
    L.call("ocr_bn_relu_bwd_apply_affine_f16", ptr(y), ptr(da), ptr(scale), ptr(shift), ptr(coef_b), ptr(coef_c),
           c_int(n), c_int(h), c_int(w), c_int(c), c_int(int(relu)), ptr(dy), c_int(0), _st())
    if L.RECORDER is not None:
        L.RECORDER.tag_last(("guest", 6.0 * n * h * w * c))           # bytes: y + da read, dy written


def bn_relu_pool_bwd_idx_apply_affine(y, argmax, da_pool, coef, relu, dy):
    n, h, w, c = y.shape
    L.call("ocr_bn_relu_pool_bwd_idx_apply_affine_f16", ptr(y), ptr(argmax), ptr(da_pool), ptr(coef[0]), ptr(coef[1]),
           ptr(coef[2]), c_int(n), c_int(h), c_int(w), c_int(c), c_int(int(relu)), ptr(dy), c_int(0), _st())
    if L.RECORDER is not None:
        L.RECORDER.tag_last(("guest", 4.75 * n * h * w * c))          # y read + dy written + the pooled gradient and index


def conv2d_wgrad(d, x, dy, dw, ws, alloc=None):
    """alloc (callable: nbytes -> uint8 tensor; layers pass Graph.empty while OCR_GUEST_BN is on): the slabs get a
    buffer of their own and the call is issued as its two halves — slab kernel ["side", FLOP], slab sum ["reduce"] —
    so that the recorded step can run the first as the host of a guest pass and the second behind the join
    (train.schedule_guests).  Same numbers either way."""
    nbytes = L.call_size("ocr_conv2d_wgrad_workspace", byref(d))
    flops = 2.0 * d.n * d.oh * d.ow * d.cout * d.cin * d.kh * d.kw
    if alloc is not None and GUEST_BN:
        buf = alloc(nbytes)
        L.call("ocr_conv2d_wgrad_slabs_f16", byref(d), ptr(x), ptr(dy), ptr(buf), c_size_t(nbytes), _st())
        if L.RECORDER is not None:
            # a host only if a guest's 56 registers per lane fit beside it (a kernel that fills the file would
            # time-slice with the guest); otherwise it stays where it was recorded
            room = L.call_int("ocr_conv2d_wgrad_guest_room", byref(d))
            L.RECORDER.tag_last(("side", flops) if room >= 56 else ("side",))
        L.call("ocr_conv2d_wgrad_reduce_f32", byref(d), ptr(buf), ptr(dw), _st())
        if L.RECORDER is not None:
            L.RECORDER.tag_last(("reduce",))
        return
    buf = ws.get(nbytes)
    L.call("ocr_conv2d_wgrad_f16", byref(d), ptr(x), ptr(dy), ptr(dw), ptr(buf), c_size_t(nbytes), _st())
    if L.RECORDER is not None:
        # (the one-call form shares the tower's slab workspace with the next weight gradient: it stays where it was
        # recorded — train.schedule_guests holds back only the split form above)
        L.RECORDER.tag_last(("side",))


def space_to_depth(x, xs):
    n, h, w, c = x.shape
    L.call("ocr_space_to_depth_f16", ptr(x), c_int(n), c_int(h), c_int(w), c_int(c), ptr(xs), _st())


def depth_to_space(xs, x, accumulate):
    n, h, w, c = x.shape
    L.call("ocr_depth_to_space_f16", ptr(xs), c_int(n), c_int(h), c_int(w), c_int(c), ptr(x), c_int(int(accumulate)), _st())


def weights_s2d(w33, w22):
    _, _, cin, cout = w33.shape
    L.call("ocr_weights_s2d_f32", ptr(w33), c_int(cin), c_int(cout), ptr(w22), _st())


def weights_s2d_grad(dw22, dw33):
    _, _, cin, cout = dw33.shape
    L.call("ocr_weights_s2d_grad_f32", ptr(dw22), c_int(cin), c_int(cout), ptr(dw33), _st())
    if L.RECORDER is not None:
        L.RECORDER.tag_last(("side",))      # reads what the weight-gradient launch before it wrote: same stream


def conv2d_first_num_mtiles(n, h, w):
    return L.call_int("ocr_conv2d_first_num_mtiles", c_int(n), c_int(h), c_int(w))


def conv2d_first(x4, w_first, y, flags=0, bias=None, stats=None, cout=None):
    """y = None with CONV_STATS: a statistics-only launch (pass `cout`)."""
    n, h, w, _ = x4.shape
    cout = y.shape[-1] if y is not None else cout
    L.call("ocr_conv2d_first_f16", c_int(n), c_int(h), c_int(w), c_int(cout), ptr(x4), ptr(w_first),
           ptr(bias), c_int(flags), ptr(y), ptr(stats), _st())


def conv2d_first_moments(x4, w_first, row, cout, ws, moments=None):
    """conv1_1's batch-norm statistics (sum y, sum y^2 per channel -> row [1][2][cout] f32, any byte buffer) from the
    image's second moments; `moments` (f64 [32][32]) keeps them for conv2d_first_wgrad_sums."""
    n, h, w, _ = x4.shape
    nbytes = L.call_size("ocr_conv2d_first_moments_workspace")
    buf = ws.get(nbytes)
    L.call("ocr_conv2d_first_moments_keep_f16", c_int(n), c_int(h), c_int(w), c_int(cout), ptr(x4), ptr(w_first),
           ptr(row), ptr(moments), ptr(buf), c_size_t(nbytes), _st())


def conv2d_first_wgrad(x4, dy, dw, ws):
    n, h, w, _ = x4.shape
    cout = dy.shape[-1]
    nbytes = L.call_size("ocr_conv2d_first_wgrad_workspace", c_int(n), c_int(h), c_int(w), c_int(cout))
    buf = ws.get(nbytes)
    L.call("ocr_conv2d_first_wgrad_f16", c_int(n), c_int(h), c_int(w), c_int(cout), ptr(x4), ptr(dy),
           ptr(dw), ptr(buf), c_size_t(nbytes), _st())


def conv2d_first_wgrad_bn(x4, da, y, shift, coef, relu, dw, ws, w_first=None):
    """conv1_1's weight gradient with its BN-backward apply computed on load (coef: bn_bwd_coefficients); with
    `w_first` y is recomputed from the image instead of read."""
    n, h, w, _ = x4.shape
    cout = da.shape[-1]
    nbytes = L.call_size("ocr_conv2d_first_wgrad_workspace", c_int(n), c_int(h), c_int(w), c_int(cout))
    buf = ws.get(nbytes)
    a, b, c = coef
    L.call("ocr_conv2d_first_wgrad_bn_f16", c_int(n), c_int(h), c_int(w), c_int(cout), ptr(x4), ptr(da),
           ptr(None if w_first is not None else y), ptr(w_first), ptr(shift), ptr(a), ptr(b), ptr(c), c_int(int(relu)),
           ptr(dw), ptr(buf), c_size_t(nbytes), _st())


def conv2d_first_bn_relu(x4, w_first, scale, shift, relu, a):
    """conv1_1's activation relu(bn(conv(x))) with the convolution evaluated again (second pass; the statistics exist)."""
    n, h, w, _ = x4.shape
    L.call("ocr_conv2d_first_bn_relu_f16", c_int(n), c_int(h), c_int(w), c_int(a.shape[-1]), ptr(x4), ptr(w_first),
           ptr(scale), ptr(shift), c_int(int(relu)), ptr(a), _st())


def conv2d_stem_num_mtiles(n, h, w):
    return L.call_int("ocr_conv2d_stem_num_mtiles", c_int(n), c_int(h), c_int(w))


def conv2d_stem(x4, w_stem, y, flags=0, bias=None, stats=None):
    n, h, w, _ = x4.shape
    L.call("ocr_conv2d_stem_f16", c_int(n), c_int(h), c_int(w), c_int(y.shape[-1]), ptr(x4), ptr(w_stem),
           ptr(bias), c_int(flags), ptr(y), ptr(stats), _st())


def conv2d_stem_wgrad(x4, dy, dw, ws):
    n, h, w, _ = x4.shape
    cout = dy.shape[-1]
    nbytes = L.call_size("ocr_conv2d_stem_wgrad_workspace", c_int(n), c_int(h), c_int(w), c_int(cout))
    buf = ws.get(nbytes)
    L.call("ocr_conv2d_stem_wgrad_f16", c_int(n), c_int(h), c_int(w), c_int(cout), ptr(x4), ptr(dy), ptr(dw),
           ptr(buf), c_size_t(nbytes), _st())


def conv2d_stem_wgrad_bn(x4, da, y, shift, coef, relu, dw, ws):
    """The root convolution's weight gradient with its BN-backward apply computed on load (coef: bn_relu_bwd_reduce)."""
    n, h, w, _ = x4.shape
    cout = da.shape[-1]
    nbytes = L.call_size("ocr_conv2d_stem_wgrad_workspace", c_int(n), c_int(h), c_int(w), c_int(cout))
    buf = ws.get(nbytes)
    a, b, c = coef
    L.call("ocr_conv2d_stem_wgrad_bn_f16", c_int(n), c_int(h), c_int(w), c_int(cout), ptr(x4), ptr(da), ptr(y),
           ptr(shift), ptr(a), ptr(b), ptr(c), c_int(int(relu)), ptr(dw), ptr(buf), c_size_t(nbytes), _st())


def bn_relu_bwd_reduce(y, scale, shift, save_mean, save_invstd, da_full, relu, dgamma, dbeta, coef, ws, da_pool=None):
    """Reduction + finalize of the BN backward, the apply step returned as coefficients (dy = A*dz + B*y + C).
    da_pool: the gradient of the layer's 2x2/2 max-pool, routed to each window's first maximum."""
    n, h, w, c = y.shape
    T = bn_bwd_num_partials(y.shape, 2 if da_pool is not None else 0)
    part, stage = ws.two(T * 2 * c * 4, bn_reduce_workspace(T, c))
    a, b, cc = coef
    L.call("ocr_bn_relu_bwd_reduce_f16", ptr(y), ptr(scale), ptr(shift), ptr(save_mean), ptr(save_invstd), ptr(da_full),
           ptr(da_pool), c_int(n), c_int(h), c_int(w), c_int(c), c_int(int(relu)), ptr(dgamma), ptr(dbeta), ptr(a), ptr(b),
           ptr(cc), ptr(part), ptr(stage), c_size_t(stage.numel()), _st())


def bn_relu_bwd_reduce_rows(y, da_full, scale, shift, save_mean, save_invstd, relu, alloc, da_pool=None, argmax=None):
    """Guest reduction pass of an end-point layer's BN backward (csrc/guest_bn.hip): returns (partial rows, T) for
    bn_bwd_coefficients_pre.  alloc: shape, dtype -> tensor (the rows outlive the call: the recorded step runs the
    finalize later)."""
    n, h, w, c = y.shape
    pooled = int(da_pool is not None)
    T = L.call_int("ocr_bn_relu_bwd_reduce_rows_count", c_int(n), c_int(h), c_int(w), c_int(c), c_int(pooled), c_int(0))
    part = alloc((T, 2, c), torch.float32)
    L.call("ocr_bn_relu_bwd_reduce_rows_f16", ptr(y), ptr(da_full), ptr(da_pool), ptr(argmax), ptr(scale), ptr(shift),
           ptr(save_mean), ptr(save_invstd), c_int(n), c_int(h), c_int(w), c_int(c), c_int(int(relu)), ptr(part), c_int(0),
           _st())
    if L.RECORDER is not None:
        # (its grid is one workgroup per CU wherever it runs: the row count must not depend on the placement)
        L.RECORDER.tag_last(("guest", (4.0 if not pooled else 4.75) * n * h * w * c, "as_is"))
    return part, T


def bn_relu_poolfull_bwd_apply_affine(y, da_full, da_pool, argmax, scale, shift, coef, relu, dy):
    """Guest apply pass of a pooled end-point layer (csrc/guest_bn.hip): dz = (da_full + routed da_pool) * ReLU mask."""
    n, h, w, c = y.shape
    L.call("ocr_bn_relu_poolfull_bwd_apply_affine_f16", ptr(y), ptr(da_full), ptr(da_pool), ptr(argmax), ptr(scale), ptr(shift),
           ptr(coef[1]), ptr(coef[2]), c_int(n), c_int(h), c_int(w), c_int(c), c_int(int(relu)), ptr(dy), c_int(0), _st())
    if L.RECORDER is not None:
        L.RECORDER.tag_last(("guest", 6.75 * n * h * w * c))          # y + da_full read, dy written, pooled gradient + index


def bn_relu_bwd_reduce_pooled(y, scale, shift, save_mean, save_invstd, pooled, relu, da_out, dgamma, dbeta, coef, ws):
    """bn_relu_bwd_reduce with the max-pool backward that produces the activation gradient folded in.
    pooled = (da_pooled [n,oh,ow,c], argmax u8, k, stride, (pad_top, pad_left)); da_out (nullable): the gathered gradient."""
    n, h, w, c = y.shape
    dap, argmax, k, stride, (pt, pl) = pooled
    T = bn_bwd_num_partials(y.shape, 0)
    part, stage = ws.two(T * 2 * c * 4, bn_reduce_workspace(T, c))
    a, b, cc = coef
    L.call("ocr_bn_relu_bwd_reduce_pooled_f16", ptr(y), ptr(scale), ptr(shift), ptr(save_mean), ptr(save_invstd), ptr(dap),
           ptr(argmax), c_int(n), c_int(h), c_int(w), c_int(c), c_int(k), c_int(stride), c_int(pt), c_int(pl),
           c_int(dap.shape[1]), c_int(dap.shape[2]), c_int(int(relu)), ptr(da_out), ptr(dgamma), ptr(dbeta), ptr(a), ptr(b),
           ptr(cc), ptr(part), ptr(stage), c_size_t(stage.numel()), _st())


def pack_weights_stem(w_hwio, w_stem):
    L.call("ocr_pack_weights_stem_f16", ptr(w_hwio), c_int(w_hwio.shape[-1]), ptr(w_stem), _st())


def pack_weights_first(w_hwio, w_first):
    L.call("ocr_pack_weights_first_f16", ptr(w_hwio), c_int(w_hwio.shape[-1]), ptr(w_first), _st())


def pack_weights(w_hwio, w_kc=None, w_ck=None):
    kh, kw, cin, cout = w_hwio.shape
    L.call("ocr_pack_weights_f16", ptr(w_hwio), c_int(kh * kw), c_int(cin), c_int(cout), ptr(w_kc),
           ptr(w_ck), _st())


class PackBatch:
    """Device table for ocr_pack_weights_batch_f16: built once for a list of (w_hwio f32, w_kc, w_ck[, ld_ck]) tensors
    (ld_ck = 32: a fuse head's [32][cin] / [cin][32] zero-padded pair; w then is the [cin][C] master)."""

    def __init__(self, entries, device):
        import ctypes
        n = len(entries)
        vp = ctypes.c_void_p * n
        ci = ctypes.c_int * n
        ent = [(e + (0,))[:4] for e in entries]
        shp = [tuple(w.shape) if w.dim() == 4 else (1, 1) + tuple(w.shape) for w, _, _, _ in ent]
        ws = vp(*[w.data_ptr() for w, _, _, _ in ent])
        kc = vp(*[(a.data_ptr() if a is not None else None) for _, a, _, _ in ent])
        ck = vp(*[(b.data_ptr() if b is not None else None) for _, _, b, _ in ent])
        taps = ci(*[s[0] * s[1] for s in shp])
        cin = ci(*[s[2] for s in shp])
        cout = ci(*[s[3] for s in shp])
        ld = ci(*[l for _, _, _, l in ent])
        nbytes = L.call_size("ocr_pack_weights_batch_table_bytes", c_int(n))
        host = torch.empty(nbytes, dtype=torch.uint8)
        grid = ctypes.c_int(0)
        L.call("ocr_pack_weights_batch_table", c_int(n), ws, taps, cin, cout, kc, ck, ld, ctypes.c_void_p(host.data_ptr()),
               ctypes.byref(grid))
        self.table = host.to(device)
        self.n, self.grid = n, grid.value
        self.keep = entries                      # the tensors the table points at

    def run(self):
        L.call("ocr_pack_weights_batch_f16", ptr(self.table), c_int(self.n), c_int(self.grid), _st())


def pack_weights_small(w, w_kc32, w_ck32):
    cin, cout = w.shape
    L.call("ocr_pack_weights_small_f16", ptr(w), c_int(cin), c_int(cout), ptr(w_kc32), ptr(w_ck32), _st())


# ------------------------------------------------------------------- image / BN / pool
def prep_images(images_f32, out_f16x4, means=(123.68, 116.78, 103.94), div=1.0):
    npix = images_f32.numel() // 3
    L.call("ocr_prep_images_norm_f16", ptr(images_f32), c_int64(npix), c_float(means[0]),
           c_float(means[1]), c_float(means[2]), c_float(div), ptr(out_f16x4), _st())


def sum_squares(x, scale, out, ws):
    """out[0] = scale * sum(x^2) (f64 accumulation, fixed order)."""
    n = x.numel()
    nbytes = L.call_size("ocr_sum_squares_workspace", c_int64(n))
    buf = ws.get(nbytes)
    L.call("ocr_sum_squares_f32", ptr(x), c_int64(n), c_float(scale), ptr(out), ptr(buf), c_size_t(nbytes), _st())


def bn_finalize(partial, T, C, count, gamma, beta, eps, decay, moving_mean, moving_var, scale, shift,
                save_mean, save_invstd, ws):
    nbytes = L.call_size("ocr_bn_reduce_workspace", c_int(T), c_int(C))
    buf = ws
    L.call("ocr_bn_finalize", ptr(partial), c_int(T), c_int(C), c_double(count), ptr(gamma), ptr(beta),
           c_float(eps), c_float(decay), ptr(moving_mean), ptr(moving_var), ptr(scale), ptr(shift),
           ptr(save_mean), ptr(save_invstd), ptr(buf), c_size_t(buf.numel() * buf.element_size()), _st())
    return nbytes


def bn_reduce_workspace(T, C):
    return L.call_size("ocr_bn_reduce_workspace", c_int(T), c_int(C))


def bn_inference_params(gamma, beta, mm, mv, eps, scale, shift):
    L.call("ocr_bn_inference_params", ptr(gamma), ptr(beta), ptr(mm), ptr(mv), c_float(eps),
           c_int(mm.numel()), ptr(scale), ptr(shift), _st())


def bn_relu(y, scale, shift, relu, pool, a_full=None, a_pool=None):
    n, h, w, c = y.shape
    L.call("ocr_bn_relu_f16", ptr(y), ptr(scale), ptr(shift), c_int(n), c_int(h), c_int(w), c_int(c),
           c_int(int(relu)), c_int(pool), ptr(a_full), ptr(a_pool), _st())


def bn_relu_pool_idx(y, scale, shift, relu, a_full, a_pool, argmax, y_pool=None):
    """bn+ReLU+2x2 max-pool that also stores the first-max position (uint8 [n,oh,ow,c]) and, with `y_pool`, the conv
    output AT that position (the BN-backward operand of the pooled positions: bn_relu_pool_bwd_idx_apply)."""
    n, h, w, c = y.shape
    L.call("ocr_bn_relu_pool_idx_f16", ptr(y), ptr(scale), ptr(shift), c_int(n), c_int(h), c_int(w), c_int(c),
           c_int(int(relu)), ptr(a_full), ptr(a_pool), ptr(argmax), ptr(y_pool), _st())


def bn_relu_pool_bwd_idx_apply(y, scale, save_mean, save_invstd, argmax, da_pool, relu, partial, T, dgamma, dbeta, dy, ws):
    """bn_relu_pool_bwd_idx without its reduction pass: `partial` [T][2][c] from the input-gradient kernel that wrote
    da_pool (conv2d_bnred with bn_y = y_pool)."""
    n, h, w, c = y.shape
    stage = ws.get(bn_reduce_workspace(T, c))
    L.call("ocr_bn_relu_pool_bwd_idx_apply_f16", ptr(y), ptr(scale), ptr(save_mean), ptr(save_invstd), ptr(argmax),
           ptr(da_pool), c_int(n), c_int(h), c_int(w), c_int(c), c_int(int(relu)), ptr(partial), c_int(T), ptr(dgamma),
           ptr(dbeta), ptr(dy), ptr(stage), c_size_t(stage.numel()), _st())


def bn_relu_pool_bwd_idx(y, scale, save_mean, save_invstd, a_pool, argmax, da_pool, relu, dgamma, dbeta, dy, ws):
    n, h, w, c = y.shape
    T = bn_bwd_num_partials(y.shape, 2)
    part, stage = ws.two(T * 2 * c * 4, bn_reduce_workspace(T, c))
    L.call("ocr_bn_relu_pool_bwd_idx_f16", ptr(y), ptr(scale), ptr(save_mean), ptr(save_invstd), ptr(a_pool),
           ptr(argmax), ptr(da_pool), c_int(n), c_int(h), c_int(w), c_int(c), c_int(int(relu)), ptr(dgamma),
           ptr(dbeta), ptr(dy), ptr(part), ptr(stage), c_size_t(stage.numel()), _st())


def bn_bwd_num_partials(shape, pool):
    n, h, w, c = shape
    return L.call_int("ocr_bn_bwd_num_partials", c_int(n), c_int(h), c_int(w), c_int(c), c_int(pool))


def bn_relu_bwd(y, scale, shift, save_mean, save_invstd, da_full, da_pool, relu, pool, dgamma, dbeta,
                dy, ws):
    n, h, w, c = y.shape
    T = bn_bwd_num_partials(y.shape, pool)
    part, stage = ws.two(T * 2 * c * 4, bn_reduce_workspace(T, c))
    L.call("ocr_bn_relu_bwd_f16", ptr(y), ptr(scale), ptr(shift), ptr(save_mean), ptr(save_invstd),
           ptr(da_full), ptr(da_pool), c_int(n), c_int(h), c_int(w), c_int(c), c_int(int(relu)),
           c_int(pool), ptr(dgamma), ptr(dbeta), ptr(dy), ptr(part), ptr(stage),
           c_size_t(stage.numel()), _st())


def channel_stats_num_partials(npix, c):
    return L.call_int("ocr_channel_stats_num_partials", c_int64(npix), c_int(c))


def channel_stats(x, partial):
    c = x.shape[-1]
    L.call("ocr_channel_stats_f16", ptr(x), c_int64(x.numel() // c), c_int(c), ptr(partial), _st())


def bn_add_relu(y, scale, shift, shortcut, out, sc_scale=None, sc_shift=None, bits=None):
    """out = relu(bn(y) + shortcut); sc_scale/sc_shift: the shortcut's own (projection) batch norm applied on the fly;
    bits: optional ReLU-mask bytes (one per 8 channels)."""
    c = y.shape[-1]
    L.call("ocr_bn_add_relu_f16", ptr(y), ptr(scale), ptr(shift), ptr(shortcut), ptr(sc_scale), ptr(sc_shift),
           c_int64(y.numel() // c), c_int(c), ptr(out), ptr(bits), _st())


def relu_bwd(out, dout, dz):
    L.call("ocr_relu_bwd_f16", ptr(out), ptr(dout), c_int64(out.numel()), ptr(dz), _st())


def add_inplace(a, b):
    L.call("ocr_add_inplace_f16", ptr(a), ptr(b), c_int64(a.numel()), _st())


def unpool_f16(x, y):
    n, lh, lw, c = x.shape
    L.call("ocr_unpool_f16", ptr(x), c_int(n), c_int(lh), c_int(lw), c_int(c), ptr(y), _st())


def unpool_bwd_f16(dy, dx, accumulate):
    n, lh, lw, c = dx.shape
    L.call("ocr_unpool_bwd_f16", ptr(dy), c_int(n), c_int(lh), c_int(lw), c_int(c), ptr(dx),
           c_int(int(accumulate)), _st())


def sc_sigmoid_split(z, c0, out0, out1):
    C = z.shape[-1]
    L.call("ocr_sc_sigmoid_split", ptr(z), c_int(z.numel() // C), c_int(C), c_int(c0), ptr(out0), ptr(out1), _st())


def sc_sigmoid_split_bwd(out0, dout0, out1, dout1, dz):
    C = dz.shape[-1]
    L.call("ocr_sc_sigmoid_split_bwd", ptr(out0), ptr(dout0), ptr(out1), ptr(dout1), c_int(dz.numel() // C), c_int(C),
           c_int(out0.shape[-1]), ptr(dz), _st())


def unpool_add_stats(t_low, y, partial):
    """y += unpool(t_low) in place; partial (nullable): [channel_stats_num_partials(y pixels, c)][2][c] f32."""
    n, lh, lw, c = t_low.shape
    L.call("ocr_unpool_add_stats_f16", ptr(t_low), c_int(n), c_int(lh), c_int(lw), c_int(c), ptr(y), ptr(partial), _st())


def sc_sigmoid(z, out):
    L.call("ocr_sc_sigmoid", ptr(z), c_int64(z.numel()), ptr(out), _st())


def sc_sigmoid_bwd(out, dout, dz):
    L.call("ocr_sc_sigmoid_bwd", ptr(out), ptr(dout), c_int64(out.numel()), ptr(dz), _st())


def bias_relu_bwd(a, da, relu, dz, dbias, ws):
    c = a.shape[-1]
    npix = a.numel() // c
    T = L.call_int("ocr_bias_relu_bwd_num_partials", c_int64(npix), c_int(c))
    buf = ws.get(T * c * 4)
    L.call("ocr_bias_relu_bwd_f16", ptr(a), ptr(da), c_int64(npix), c_int(c), c_int(int(relu)), ptr(dz),
           ptr(dbias), ptr(buf), _st())


def sc_colsum(x, C, out, ws):
    P = x.numel() // C
    T = sc_num_partials(P, C)
    buf = ws.get((T + 1) * 2 * C * 4)
    L.call("ocr_sc_colsum", ptr(x), c_int(P), c_int(C), ptr(out), ptr(buf), _st())


def maxpool(x, k, stride, pad, y, argmax=None):
    n, h, w, c = x.shape
    _, oh, ow, _ = y.shape
    L.call("ocr_maxpool_f16", ptr(x), c_int(n), c_int(h), c_int(w), c_int(c), c_int(k), c_int(stride),
           c_int(pad[0]), c_int(pad[1]), c_int(oh), c_int(ow), ptr(y), ptr(argmax), _st())


def bn_relu_maxpool(bn_y, scale, shift, relu, k, stride, pad, y, argmax=None):
    """max-pool over relu(bn(bn_y)) computed on the fly from the raw conv output."""
    n, h, w, c = bn_y.shape
    _, oh, ow, _ = y.shape
    L.call("ocr_bn_relu_maxpool_f16", ptr(bn_y), ptr(scale), ptr(shift), c_int(int(relu)), c_int(n), c_int(h), c_int(w),
           c_int(c), c_int(k), c_int(stride), c_int(pad[0]), c_int(pad[1]), c_int(oh), c_int(ow), ptr(y), ptr(argmax), _st())


def maxpool_bwd(x, dy, k, stride, pad, dx, accumulate, argmax=None, in_shape=None):
    """Either `x` (arg-max re-derived) or the forward pass's `argmax` index tensor."""
    n, h, w, c = in_shape if in_shape is not None else x.shape
    _, oh, ow, _ = dy.shape
    L.call("ocr_maxpool_bwd_f16", ptr(None if argmax is not None else x), ptr(argmax), ptr(dy), c_int(n),
           c_int(h), c_int(w), c_int(c), c_int(k), c_int(stride), c_int(pad[0]), c_int(pad[1]), c_int(oh),
           c_int(ow), ptr(dx), c_int(int(accumulate)), _st())


# ------------------------------------------------------------------------------ heads
def conv1x1_small(x, w_kc32, cout, out, bias=None):
    P = x.numel() // x.shape[-1]
    L.call("ocr_conv1x1_small_f16", ptr(x), ptr(w_kc32), ptr(bias), c_int(P), c_int(x.shape[-1]),
           c_int(cout), ptr(out), _st())


def conv1x1_small_dgrad(dz, w_ck32, cout, dx, accumulate, grad_scale=1.0):
    P = dx.numel() // dx.shape[-1]
    L.call("ocr_conv1x1_small_dgrad_f16", ptr(dz), ptr(w_ck32), c_int(P), c_int(dx.shape[-1]),
           c_int(cout), c_float(grad_scale), ptr(dx), c_int(int(accumulate)), _st())


def conv1x1_small_wgrad(x, dz, cout, dw, ws):
    cin = x.shape[-1]
    P = x.numel() // cin
    nbytes = L.call_size("ocr_conv1x1_small_wgrad_workspace", c_int(P), c_int(cin), c_int(cout))
    buf = ws.get(nbytes)
    L.call("ocr_conv1x1_small_wgrad_f16", ptr(x), ptr(dz), c_int(P), c_int(cin), c_int(cout), ptr(dw),
           ptr(buf), c_size_t(nbytes), _st())


def sc_num_partials(P, C):
    return L.call_int("ocr_sc_num_partials", c_int(P), c_int(C))


def sc_stats(x, C, partial):
    P = x.numel() // C
    L.call("ocr_sc_stats", ptr(x), c_int(P), c_int(C), ptr(partial), _st())


def sc_fuse(out, za=None, sa=None, ha=None, zb=None, sb=None, hb=None, prev=None, relu=True):
    n, h, w, C = out.shape
    L.call("ocr_sc_fuse", ptr(za), ptr(sa), ptr(ha), ptr(zb), ptr(sb), ptr(hb), ptr(prev), c_int(n),
           c_int(h), c_int(w), c_int(C), c_int(int(relu)), ptr(out), _st())


def sc_unpool_bwd(dout, dprev):
    n, lh, lw, C = dprev.shape
    L.call("ocr_sc_unpool_bwd", ptr(dout), c_int(n), c_int(lh), c_int(lw), c_int(C), ptr(dprev), _st())


def sc_bn_bwd(z, scale, shift, save_mean, save_invstd, dout, relu, dgamma, dbeta, dz, ws):
    C = z.shape[-1]
    P = z.numel() // C
    T = sc_num_partials(P, C)
    buf = ws.get((T + 1) * 2 * C * 4)
    L.call("ocr_sc_bn_bwd", ptr(z), ptr(scale), ptr(shift), ptr(save_mean), ptr(save_invstd), ptr(dout),
           c_int(P), c_int(C), c_int(int(relu)), ptr(dgamma), ptr(dbeta), ptr(dz), ptr(buf), _st())


def sc_pointwise_fwd(x, xo, cin, w, out, oo, cout, bias=None):
    ldx, ldo = x.shape[-1], out.shape[-1]
    P = x.numel() // ldx
    L.call("ocr_sc_pointwise_fwd", ptr(x), c_int(ldx), c_int(xo), c_int(cin), ptr(w), ptr(bias), c_int(P),
           ptr(out), c_int(ldo), c_int(oo), c_int(cout), _st())


def sc_pointwise_dgrad(dout, oo, cout, w, dx, xo, cin):
    ldx, ldo = dx.shape[-1], dout.shape[-1]
    P = dx.numel() // ldx
    L.call("ocr_sc_pointwise_dgrad", ptr(dout), c_int(ldo), c_int(oo), c_int(cout), ptr(w), c_int(P),
           ptr(dx), c_int(ldx), c_int(xo), c_int(cin), _st())


def sc_pointwise_wgrad(x, xo, cin, dout, oo, cout, dw, db, ws):
    ldx, ldo = x.shape[-1], dout.shape[-1]
    P = x.numel() // ldx
    nbytes = L.call_size("ocr_sc_pointwise_wgrad_workspace", c_int(cin), c_int(cout))
    buf = ws.get(nbytes)
    L.call("ocr_sc_pointwise_wgrad", ptr(x), c_int(ldx), c_int(xo), c_int(cin), ptr(dout), c_int(ldo),
           c_int(oo), c_int(cout), c_int(P), ptr(dw), ptr(db), ptr(buf), c_size_t(nbytes), _st())


# ------------------------------------------------------------------------------- loss
def dice_loss_fwd(yt_pixel, yp_pixel, yt_link, yp_link, mask, sums27, loss10, ws):
    P = mask.numel()
    pc = yp_pixel.numel() // P
    G = yp_link.numel() // (8 * P)
    nbytes = L.call_size("ocr_dice_workspace", c_int(P))
    buf = ws.get(nbytes)
    L.call("ocr_dice_loss_fwd", ptr(yt_pixel), ptr(yp_pixel), c_int(pc), ptr(yt_link), ptr(yp_link),
           c_int(G), ptr(mask), c_int(P), ptr(sums27), ptr(loss10), ptr(buf), c_size_t(nbytes), _st())


def dice_loss_bwd(yt_pixel, yt_link, mask, sums27, grad_scale, d_pixel, d_link):
    P = mask.numel()
    pc = d_pixel.numel() // P
    G = d_link.numel() // (8 * P)
    L.call("ocr_dice_loss_bwd", ptr(yt_pixel), c_int(pc), ptr(yt_link), c_int(G), ptr(mask), c_int(P),
           ptr(sums27), c_float(grad_scale), ptr(d_pixel), ptr(d_link), _st())


def softmax_loss_fwd(desc, pixel_logits, link_logits, pixel_labels, link_labels, thr, sums34, loss10, ws):
    nbytes = L.call_size("ocr_softmax_loss_workspace", byref(desc))
    buf = ws.get(nbytes)
    L.call("ocr_softmax_loss_fwd", byref(desc), ptr(pixel_logits), ptr(link_logits), ptr(pixel_labels),
           ptr(link_labels), ptr(thr), ptr(sums34), ptr(loss10), ptr(buf), c_size_t(nbytes), _st())


def softmax_loss_selected(desc, pixel_logits, pixel_labels, thr, mask_u8):
    L.call("ocr_softmax_loss_selected", byref(desc), ptr(pixel_logits), ptr(pixel_labels), ptr(thr), ptr(mask_u8), _st())


def softmax_loss_bwd(desc, pixel_logits, link_logits, pixel_labels, link_labels, thr, sums34, grad_scale,
                     d_pixel, d_link):
    L.call("ocr_softmax_loss_bwd", byref(desc), ptr(pixel_logits), ptr(link_logits), ptr(pixel_labels),
           ptr(link_labels), ptr(thr), ptr(sums34), c_float(grad_scale), ptr(d_pixel), ptr(d_link), _st())


# ------------------------------------------------------------------ heads, batched (one launch per kernel kind)
def _struct(name, fields):
    return type(name, (ctypes.Structure,), {"_fields_": fields})


_VP, _I32 = ctypes.c_void_p, ctypes.c_int32
HeadConvItem = _struct("HeadConvItem", [("x", _VP), ("w_kc32", _VP), ("bias", _VP), ("out", _VP), ("stats_partial", _VP),
                                        ("P", _I32), ("cin", _I32), ("cout", _I32)])
HeadDgradItem = _struct("HeadDgradItem", [("dz", _VP), ("w_ck32", _VP), ("dx", _VP), ("P", _I32), ("cin", _I32),
                                          ("cout", _I32), ("accumulate", _I32)])
HeadWgradItem = _struct("HeadWgradItem", [("x", _VP), ("dz", _VP), ("dw", _VP), ("slab", _VP), ("P", _I32), ("cin", _I32),
                                          ("cout", _I32)])
BnFinalizeItem = _struct("BnFinalizeItem", [("partial", _VP), ("T", _I32), ("C", _I32), ("count", c_double), ("gamma", _VP),
                                            ("beta", _VP), ("moving_mean", _VP), ("moving_var", _VP), ("scale", _VP),
                                            ("shift", _VP), ("save_mean", _VP), ("save_invstd", _VP)])
ScBnBwdItem = _struct("ScBnBwdItem", [("z", _VP), ("scale", _VP), ("shift", _VP), ("save_mean", _VP), ("save_invstd", _VP),
                                      ("dout", _VP), ("dgamma", _VP), ("dbeta", _VP), ("dz", _VP), ("partial", _VP),
                                      ("P", _I32), ("C", _I32), ("relu", _I32)])
ScColsumItem = _struct("ScColsumItem", [("x", _VP), ("out", _VP), ("partial", _VP), ("P", _I32), ("C", _I32)])
ScActItem = _struct("ScActItem", [("z", _VP), ("scale", _VP), ("shift", _VP), ("out", _VP), ("total", c_int64), ("C", _I32)])


def _dp(t):
    return t.data_ptr() if t is not None else None


def _arr(cls, rows):
    return (cls * len(rows))(*[cls(*r) for r in rows])


def conv1x1_small_batch_rows(P):
    return L.call_int("ocr_conv1x1_small_batch_rows", c_int(P))


def conv1x1_small_batch(items):
    """items: [(x f16 [P,cin], w_kc32, bias | None, out f32 [P,cout], stats_partial | None)]"""
    rows = [(_dp(x), _dp(w), _dp(b), _dp(o), _dp(sp), x.numel() // x.shape[-1], x.shape[-1], o.shape[-1])
            for x, w, b, o, sp in items]
    L.call("ocr_conv1x1_small_batch_f16", _arr(HeadConvItem, rows), c_int(len(rows)), _st())


def conv1x1_small_dgrad_batch(items, grad_scale=1.0):
    """items: [(dz f32 [P,cout], w_ck32, dx f16 [P,cin], accumulate)]"""
    rows = [(_dp(dz), _dp(w), _dp(dx), dx.numel() // dx.shape[-1], dx.shape[-1], dz.shape[-1], int(acc))
            for dz, w, dx, acc in items]
    L.call("ocr_conv1x1_small_dgrad_batch_f16", _arr(HeadDgradItem, rows), c_int(len(rows)), c_float(grad_scale), _st())


def conv1x1_small_wgrad_batch_slab_bytes(P, cin):
    return L.call_size("ocr_conv1x1_small_wgrad_batch_slab_bytes", c_int(P), c_int(cin))


def conv1x1_small_wgrad_batch(items):
    """items: [(x f16 [P,cin], dz f32 [P,cout], dw f32 [cin,cout], slab)]"""
    rows = [(_dp(x), _dp(dz), _dp(dw), _dp(slab), x.numel() // x.shape[-1], x.shape[-1], dz.shape[-1])
            for x, dz, dw, slab in items]
    L.call("ocr_conv1x1_small_wgrad_batch_f16", _arr(HeadWgradItem, rows), c_int(len(rows)), _st())


def bn_finalize_batch(items, eps, decay):
    """items: [(partial, T, C, count, gamma, beta, moving_mean, moving_var, scale, shift, save_mean, save_invstd)]"""
    rows = [(_dp(p), T, C, float(cnt), _dp(ga), _dp(be), _dp(mm), _dp(mv), _dp(sc), _dp(sh), _dp(mu), _dp(istd))
            for p, T, C, cnt, ga, be, mm, mv, sc, sh, mu, istd in items]
    L.call("ocr_bn_finalize_batch", _arr(BnFinalizeItem, rows), c_int(len(rows)), c_float(eps), c_float(decay), _st())


def sc_bn_bwd_batch(items):
    """items: [(z, scale, shift, save_mean, save_invstd, dout, dgamma, dbeta, dz, partial, relu)] on [P][C] f32 tensors"""
    rows = [(_dp(z), _dp(sc), _dp(sh), _dp(mu), _dp(istd), _dp(do), _dp(dg), _dp(db), _dp(dz), _dp(part),
             z.numel() // z.shape[-1], z.shape[-1], int(relu))
            for z, sc, sh, mu, istd, do, dg, db, dz, part, relu in items]
    L.call("ocr_sc_bn_bwd_batch", _arr(ScBnBwdItem, rows), c_int(len(rows)), _st())


def sc_colsum_batch(items):
    """items: [(x f32 [P,C], out f32 [C], partial)]"""
    rows = [(_dp(x), _dp(o), _dp(part), x.numel() // x.shape[-1], x.shape[-1]) for x, o, part in items]
    L.call("ocr_sc_colsum_batch", _arr(ScColsumItem, rows), c_int(len(rows)), _st())


def sc_act_batch(items, relu):
    """items: [(z, scale, shift, out)]"""
    rows = [(_dp(z), _dp(sc), _dp(sh), _dp(o), z.numel(), z.shape[-1]) for z, sc, sh, o in items]
    L.call("ocr_sc_act_batch", _arr(ScActItem, rows), c_int(len(rows)), c_int(int(relu)), _st())


def sc_pointwise_pair_num_partials(P):
    return L.call_int("ocr_sc_pointwise_pair_num_partials", c_int(P))


def sc_pointwise_pair_fwd(x18, w_px, b_px, w_lk, b_lk, z_px, z_lk, part_px=None, part_lk=None):
    P = x18.numel() // 18
    L.call("ocr_sc_pointwise_pair_fwd", ptr(x18), ptr(w_px), ptr(b_px), ptr(w_lk), ptr(b_lk), c_int(P), ptr(z_px),
           ptr(z_lk), ptr(part_px), ptr(part_lk), _st())


def sc_pointwise_pair_bwd(x18, dz_px, dz_lk, w_px, w_lk, dx18, dw_px, db_px, dw_lk, db_lk, ws):
    P = x18.numel() // 18
    nbytes = L.call_size("ocr_sc_pointwise_pair_bwd_workspace")
    buf = ws.get(nbytes)
    L.call("ocr_sc_pointwise_pair_bwd", ptr(x18), ptr(dz_px), ptr(dz_lk), ptr(w_px), ptr(w_lk), c_int(P), ptr(dx18),
           ptr(dw_px), ptr(db_px), ptr(dw_lk), ptr(db_lk), ptr(buf), c_size_t(nbytes), _st())


def label_masks(labels, label_rule, pos_u8, neg_u8):
    L.call("ocr_label_masks", ptr(labels), c_int64(labels.numel()), c_int(label_rule), ptr(pos_u8), ptr(neg_u8), _st())


def ohnm_select(scores, pos_u8, neg_u8, n_pos_i32, n, hw, neg_ratio, selected_neg, selected):
    L.call("ocr_ohnm_select", ptr(scores), ptr(pos_u8), ptr(neg_u8), ptr(n_pos_i32), c_int(n), c_int(hw),
           c_float(neg_ratio), ptr(selected_neg), ptr(selected), _st())


def link_ce_fwd(gt, gt_stride, pred, pred_stride, w_pixel, count, sums4, loss1, ws):
    nbytes = L.call_size("ocr_link_ce_workspace", c_int64(count))
    buf = ws.get(nbytes)
    L.call("ocr_link_ce_fwd", ptr(gt), c_int(gt_stride), ptr(pred), c_int(pred_stride), ptr(w_pixel), c_int64(count),
           ptr(sums4), ptr(loss1), ptr(buf), c_size_t(nbytes), _st())


def link_ce_bwd(gt, gt_stride, pred, pred_stride, w_pixel, count, sums4, grad_scale, d_pred, d_stride):
    L.call("ocr_link_ce_bwd", ptr(gt), c_int(gt_stride), ptr(pred), c_int(pred_stride), ptr(w_pixel), c_int64(count),
           ptr(sums4), c_float(grad_scale), ptr(d_pred), c_int(d_stride), _st())


# ----------------------------------------------------------------------------- decode
def softmax_pairs(logits, probs):
    L.call("ocr_softmax_pairs", ptr(logits), c_int64(logits.numel() // 2), ptr(probs), _st())


def link_softmax_stack(link_logits, out):
    L.call("ocr_link_softmax_stack", ptr(link_logits), c_int64(link_logits.numel() // 16), ptr(out), _st())


def pixel_detect(score_map, link_scores, n, h, w, score_thresh, link_thresh, mask):
    L.call("ocr_pixel_detect", ptr(score_map), ptr(link_scores), c_int(n), c_int(h), c_int(w),
           c_float(score_thresh), c_float(link_thresh), ptr(mask), _st())


def link_cc(pixel_score, link_score, stride, offset, n, h, w, pixel_thresh, link_thresh, min_size, labels,
            ncomp, comps, ws):
    nbytes = L.call_size("ocr_link_cc_workspace", c_int(n), c_int(h), c_int(w))
    buf = ws.get(nbytes)
    L.call("ocr_link_cc", ptr(pixel_score), ptr(link_score), c_int(stride), c_int(offset), c_int(n), c_int(h),
           c_int(w), c_float(pixel_thresh), c_float(link_thresh), c_int(min_size), ptr(labels), ptr(ncomp),
           ptr(comps), c_int(comps.shape[1]), ptr(buf), c_size_t(nbytes), _st())


def link_cc_directed(pixel_score, link_score, stride, offset, n, h, w, pixel_thresh, link_thresh, min_size,
                     union_labels, union_ncomp, labels, ncomp, comps, ws, seed_order=None):
    """The reference's directed-DFS grouping, refining ocr_link_cc's components (include/ocr_hip.h).
    seed_order: int32 [n, h*w] device tensor from `py27_dict_order` (None: ascending pixel index)."""
    nbytes = L.call_size("ocr_link_cc_directed_workspace", c_int(n), c_int(h), c_int(w))
    buf = ws.get(nbytes)
    L.call("ocr_link_cc_directed", ptr(pixel_score), ptr(link_score), c_int(stride), c_int(offset), c_int(n),
           c_int(h), c_int(w), c_float(pixel_thresh), c_float(link_thresh), c_int(min_size), ptr(union_labels),
           ptr(union_ncomp), ptr(seed_order), ptr(labels), ptr(ncomp), ptr(comps), c_int(comps.shape[1]), ptr(buf),
           c_size_t(nbytes), _st())


def py27_dict_order(pixel_score_host, pixel_thresh):
    """HOST routine: pixel scores f32 [n,h,w] CPU tensor -> int32 [n, h*w] CPU tensor, per image the keys of the script's
    Python-2 dict in `graph.keys()` order, -1 padded (ocr_py27_dict_order)."""
    t = pixel_score_host
    assert t.dtype == torch.float32 and not t.is_cuda and t.is_contiguous()
    n, h, w = t.shape
    out = torch.empty((n, h * w), dtype=torch.int32)
    fn = L._fn("ocr_py27_dict_order", c_int)
    for b in range(n):
        rc = fn(ctypes.c_void_p(t[b].data_ptr()), c_float(pixel_thresh), c_int(h), c_int(w), ctypes.c_void_p(out[b].data_ptr()))
        if rc < 0:
            L.check(rc, "ocr_py27_dict_order")
    return out


def lanms(boxes, counts, iou_thresh, merged, n_merged, keep_idx, n_keep, ws):
    n_images, max_k, _ = boxes.shape
    nbytes = L.call_size("ocr_lanms_workspace", c_int(n_images), c_int(max_k))
    buf = ws.get(nbytes)
    L.call("ocr_lanms", ptr(boxes), ptr(counts), c_int(n_images), c_int(max_k), c_float(iou_thresh),
           ptr(merged), ptr(n_merged), ptr(keep_idx), ptr(n_keep), ptr(buf), c_size_t(nbytes), _st())


def min_area_rects(labels, ncomp, max_comps, scale_x, scale_y, hull_n, hull_head, calipers, ws):
    n, h, w = labels.shape
    nbytes = L.call_size("ocr_min_area_rects_workspace", c_int(n), c_int(h), c_int(w), c_int(max_comps))
    buf = ws.get(nbytes)
    L.call("ocr_min_area_rects", ptr(labels), ptr(ncomp), c_int(n), c_int(h), c_int(w), c_int(max_comps),
           c_double(scale_x), c_double(scale_y), ptr(hull_n), ptr(hull_head), ptr(calipers), ptr(buf),
           c_size_t(nbytes), _st())


def mask_cc(mask, value, connectivity, labels, ncomp, comps, ws):
    n, h, w = mask.shape
    nbytes = L.call_size("ocr_link_cc_workspace", c_int(n), c_int(h), c_int(w))
    buf = ws.get(nbytes)
    L.call("ocr_mask_cc", ptr(mask), c_int(value), c_int(connectivity), c_int(n), c_int(h), c_int(w), ptr(labels),
           ptr(ncomp), ptr(comps), c_int(comps.shape[1]), ptr(buf), c_size_t(nbytes), _st())


def hole_border_rects(mask, zlabels, nregions, max_regions, scale_x, scale_y, hull_n, hull_head, calipers, ws):
    n, h, w = mask.shape
    nbytes = L.call_size("ocr_hole_border_rects_workspace", c_int(n), c_int(h), c_int(w), c_int(max_regions))
    buf = ws.get(nbytes)
    L.call("ocr_hole_border_rects", ptr(mask), ptr(zlabels), ptr(nregions), c_int(n), c_int(h), c_int(w),
           c_int(max_regions), c_double(scale_x), c_double(scale_y), ptr(hull_n), ptr(hull_head), ptr(calipers),
           ptr(buf), c_size_t(nbytes), _st())


def east_pixel_detect(score, link16, h, w, score_thresh, link_thresh, mask, first_second):
    L.call("ocr_east_pixel_detect", ptr(score), ptr(link16), c_int(h), c_int(w), c_float(score_thresh),
           c_float(link_thresh), ptr(mask), ptr(first_second), _st())


def contour_parents(labels, zlabels, comps, ncomp, zcomps, nregions, parent_c, parent_z):
    h, w = labels.shape[-2:]
    L.call("ocr_contour_parents", ptr(labels), ptr(zlabels), ptr(comps), c_int(ncomp), ptr(zcomps), c_int(nregions),
           c_int(h), c_int(w), ptr(parent_c), ptr(parent_z), _st())


def zero_pixels(mask, idx):
    L.call("ocr_zero_pixels_u8", ptr(mask), ptr(idx), c_int(idx.numel()), _st())


# ----------------------------------------------------------------------------- labels
def poly_cover(polys, counts, ignore, h, w, cover):
    n, P, V, _ = polys.shape
    L.call("ocr_poly_cover", ptr(polys), ptr(counts), ptr(ignore), c_int(n), c_int(P), c_int(V), c_int(h),
           c_int(w), ptr(cover), _st())


def icdar_labels(cover, step, score, geo, mask):
    n, h, w = cover.shape
    L.call("ocr_icdar_labels", ptr(cover), c_int(n), c_int(h), c_int(w), c_int(step), ptr(score), ptr(geo),
           ptr(mask), _st())


def pixellink_labels(cover, new_h, new_w, score, link):
    n, h, w = cover.shape
    L.call("ocr_pixellink_labels", ptr(cover), c_int(n), c_int(h), c_int(w), c_int(new_h), c_int(new_w),
           ptr(score), ptr(link), _st())


def resize_linear_u8(src_u8, dst_f32):
    H, W, cn = src_u8.shape
    dh, dw, _ = dst_f32.shape
    L.call("ocr_resize_linear_u8", ptr(src_u8), c_int(H), c_int(W), c_int(cn), ptr(dst_f32), c_int(dh),
           c_int(dw), _st())


def resize_cubic_f32(src_f32, dst_f32, pre_scale=1.0, post_scale=1.0):
    """src f32 [planes,h,w] -> dst f32 [planes,dh,dw] (cv2.resize INTER_CUBIC per plane)."""
    planes, h, w = src_f32.shape
    p2, dh, dw = dst_f32.shape
    assert planes == p2 and src_f32.dtype == torch.float32 and dst_f32.dtype == torch.float32
    L.call("ocr_resize_cubic_f32", ptr(src_f32), c_int(planes), c_int(h), c_int(w), ptr(dst_f32), c_int(dh),
           c_int(dw), c_float(pre_scale), c_float(post_scale), _st())


def quad_iou(dets, gts, mask_h, mask_w, inter, uni):
    nd, V, _ = dets.shape
    ng = gts.shape[0]
    L.call("ocr_quad_iou", ptr(dets), c_int(nd), ptr(gts), c_int(ng), c_int(V), c_int(mask_h), c_int(mask_w),
           ptr(inter), ptr(uni), _st())


# -------------------------------------------------------------------------- optimiser
def adam_step(w, g, m, v, ema, n_reg, lr_t, beta1, beta2, eps, wd, inv_scale, ema_decay):
    L.call("ocr_adam_step", ptr(w), ptr(g), ptr(m), ptr(v), ptr(ema), c_int64(w.numel()),
           c_int64(n_reg), c_float(lr_t), c_float(beta1), c_float(beta2), c_float(eps), c_float(wd),
           c_float(inv_scale), c_float(ema_decay), _st())


def momentum_step(w, g, acc, ema, n_reg, lr, momentum, wd, inv_scale, ema_decay):
    L.call("ocr_momentum_step", ptr(w), ptr(g), ptr(acc), ptr(ema), c_int64(w.numel()), c_int64(n_reg),
           c_float(lr), c_float(momentum), c_float(wd), c_float(inv_scale), c_float(ema_decay), _st())


def scale_(x, s):
    L.call("ocr_scale_f32", ptr(x), c_int64(x.numel()), c_float(s), _st())


def fill_(x, value=0.0):
    """Fill a 4-byte-element tensor (or an even-length f16 tensor, value 0 only) with `value`."""
    nbytes = x.numel() * x.element_size()
    assert nbytes % 4 == 0 and (x.element_size() == 4 or value == 0.0)
    L.call("ocr_fill_f32", ptr(x), c_int64(nbytes // 4), c_float(value), _st())


# ------------------------------------------------------------------- f32 inference precision
# (product library: matrix-core f32 convolution + element-wise f32 kernels, csrc/f32_infer.hip; the plain direct
#  convolution of libocr_verify.so / include/ocr_verify.h is its independent checker: F32_CONV = "direct")
F32_CONV = __import__("os").environ.get("OCR_F32_CONV", "mfma")


def conv2d_f32(d, x, w_hwio, y, bias=None, route=None):
    if (route or F32_CONV) == "direct":
        L.call_verify("ocr_conv2d_f32", byref(d), ptr(x), ptr(w_hwio), ptr(bias), ptr(y), _st())
    else:
        L.call("ocr_conv2d_f32_mfma", byref(d), ptr(x), ptr(w_hwio), ptr(bias), ptr(y), _st())


def channel_stats_f32_num_partials(npix, c):
    return L.call_int("ocr_channel_stats_f32_num_partials", c_int64(npix), c_int(c))


def channel_stats_f32(x, c, partial):
    L.call("ocr_channel_stats_f32", ptr(x), c_int64(x.numel() // c), c_int(c), ptr(partial), _st())


def bn_relu_f32(y, scale, shift, relu, pool, a_full=None, a_pool=None):
    n, h, w, c = y.shape
    L.call("ocr_bn_relu_f32", ptr(y), ptr(scale), ptr(shift), c_int(n), c_int(h), c_int(w), c_int(c),
           c_int(int(relu)), c_int(pool), ptr(a_full), ptr(a_pool), _st())


def maxpool_f32(x, k, stride, pad, y):
    n, h, w, c = x.shape
    _, oh, ow, _ = y.shape
    L.call("ocr_maxpool_f32", ptr(x), c_int(n), c_int(h), c_int(w), c_int(c), c_int(k), c_int(stride),
           c_int(pad[0]), c_int(pad[1]), c_int(oh), c_int(ow), ptr(y), _st())


def prep_images_f32(images, out, means):
    L.call("ocr_prep_images_f32", ptr(images), c_int64(images.numel() // 3), c_float(means[0]),
           c_float(means[1]), c_float(means[2]), ptr(out), _st())


def bn_add_relu_f32(y, scale, shift, shortcut, out):
    c = y.shape[-1]
    L.call("ocr_bn_add_relu_f32", ptr(y), ptr(scale), ptr(shift), ptr(shortcut), c_int64(y.numel() // c),
           c_int(c), ptr(out), _st())


def unpool_f32(x, y):
    n, h, w, c = x.shape
    L.call("ocr_unpool_f32", ptr(x), c_int(n), c_int(h), c_int(w), c_int(c), ptr(y), _st())
