"""ctypes binding of libocr_hip.so (C ABI in include/ocr_hip.h).

The product path has no CPU fallback: if the HIP library is missing, importing
this module still works (so CPU-only host logic is testable) but any kernel call
raises `OcrHipError` loudly.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# 16-bit storage type of activations / packed weights, process wide: "f16" (default) or "bf16"
# (OCR_STORAGE=bf16 selects libocr_hip_bf16.so, the same sources built with bfloat16 storage + bf16 MFMA)
STORAGE = os.environ.get("OCR_STORAGE", "f16")
if STORAGE not in ("f16", "bf16"):
    raise ValueError("OCR_STORAGE must be f16 or bf16, got %r" % STORAGE)
LIB_PATH = os.environ.get("OCR_HIP_LIB", os.path.join(_HERE, "libocr_hip.so" if STORAGE == "f16" else "libocr_hip_bf16.so"))


class OcrHipError(RuntimeError):
    pass


class ConvDesc(ctypes.Structure):
    """ocr_conv_desc (include/ocr_hip.h)."""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "n", "h", "w", "cin", "oh", "ow", "cout", "kh", "kw", "stride",
        "dilation", "pad_top", "pad_left", "flip_taps", "flags")]


CONV_BIAS, CONV_RELU, CONV_STATS, CONV_ACCUM_F16 = 1, 2, 4, 8


class SoftmaxLossDesc(ctypes.Structure):
    """ocr_softmax_loss_desc (include/ocr_hip.h)."""
    _fields_ = [(n, ctypes.c_int32) for n in ("n", "hw", "pixel_rule", "label_rule", "link_gate", "focal")] + \
               [(n, ctypes.c_float) for n in ("neg_ratio", "alpha", "gamma")]

_lib = None
# the value of OCR_ABI_VERSION (include/ocr_hip.h) this host layer was written against: a library built from
# other sources would take its pointers shifted by a slot and write wildly instead of returning an OCR_ERR
ABI_VERSION = 7


def load():
    """dlopen the in-tree library; raises OcrHipError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own HIP runtime; import it first so libocr_hip.so binds to the SAME
    # libamdhip64 (streams and device pointers come from torch's allocator).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise OcrHipError(
            "libocr_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C tensorflow_ocr_amd/csrc` (no CPU fallback exists for the product path)")
    lib = ctypes.CDLL(LIB_PATH)
    lib.ocr_status_string.restype = ctypes.c_char_p
    lib.ocr_status_string.argtypes = [ctypes.c_int]
    lib.ocr_abi_version.restype = ctypes.c_int
    lib.ocr_storage_dtype.restype = ctypes.c_char_p
    if lib.ocr_abi_version() != ABI_VERSION:
        raise OcrHipError("%s exports ABI version %d, the Python host expects %d: rebuild it (`make -C "
                          "tensorflow_ocr_amd/csrc`)" % (LIB_PATH, lib.ocr_abi_version(), ABI_VERSION))
    if lib.ocr_storage_dtype().decode() != STORAGE:
        raise OcrHipError("%s stores %s but OCR_STORAGE=%s" % (LIB_PATH, lib.ocr_storage_dtype().decode(), STORAGE))
    _lib = lib
    return lib


VERIFY_LIB_PATH = os.path.join(_HERE, "libocr_verify.so")
_verify = None


def load_verify():
    """libocr_verify.so (include/ocr_verify.h): the f32 VERIFICATION kernels.  Test infrastructure —
    only `Graph(precision="f32")` (layers_f32.py) reaches it; the product path never does."""
    global _verify
    if _verify is None:
        import torch  # noqa: F401
        if not os.path.exists(VERIFY_LIB_PATH):
            raise OcrHipError("libocr_verify.so is not built: run `make -C tensorflow_ocr_amd/csrc`")
        _verify = ctypes.CDLL(VERIFY_LIB_PATH)
    return _verify


def call_verify(name, *args, restype=ctypes.c_int):
    fn = getattr(load_verify(), name)
    fn.restype = restype
    rc = fn(*args)
    if restype is ctypes.c_int and rc < 0:
        check(rc, name)
    return rc


def check(status, what=""):
    if status != 0:
        lib = load()
        msg = lib.ocr_status_string(int(status)).decode()
        raise OcrHipError("%s failed: %s (%d)" % (what or "ocr call", msg, status))


class Recorder:
    """Records the flat sequence of C-ABI calls (and host callbacks) of one training step so that
    later steps replay it without re-running the Python graph logic: the step's shapes, buffers
    and launch order are static, so the recorded (function, ctypes arguments) pairs stay valid as
    long as every tensor they point to is kept alive (graph.Graph.keepalive)."""

    def __init__(self):
        self.entries = []     # ["c", fn, args, name, tag] | ["py", callable]
        # only the thread that created the recorder is recorded (the feeder thread launches its own
        # resize / label kernels concurrently); the owner is part of the object, so it is set before
        # the recorder can be seen through `RECORDER`
        self.owner = threading.get_ident()

    def mine(self):
        return self.owner == threading.get_ident()

    def c(self, fn, args, name):
        self.entries.append(["c", fn, args, name, None])

    def py(self, fn):
        if self.mine():
            self.entries.append(["py", fn])

    def tag_last(self, tag):
        if self.mine():
            self.entries[-1][4] = tag


RECORDER = None
_fn_cache = {}
_STREAM = None


def set_stream(handle):
    """Pin the stream pointer handed to every call (None: query torch each time)."""
    global _STREAM
    _STREAM = handle


def ptr(t):
    """Device pointer of a torch tensor (or None -> NULL)."""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    if _STREAM is not None:
        return _STREAM
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _fn(name, restype):
    key = (name, restype)
    fn = _fn_cache.get(key)
    if fn is None:
        fn = getattr(load(), name)
        fn.restype = restype
        _fn_cache[key] = fn
    return fn


def call(name, *args):
    """Call `ocr_<name>` and raise on a non-zero status."""
    fn = _fn(name, ctypes.c_int)
    rc = fn(*args)
    if rc != 0:
        check(rc, name)
    rec = RECORDER
    if rec is not None and rec.mine():
        rec.c(fn, args, name)
    return rc


def call_size(name, *args):
    """Call a `size_t`-returning query."""
    return int(_fn(name, ctypes.c_size_t)(*args))


def call_int(name, *args):
    v = int(_fn(name, ctypes.c_int)(*args))
    if v < 0:
        check(v, name)
    return v


def csrc_fingerprint():
    """sha256 (first 16 hex digits) over the kernel sources — csrc/*.hip, *.h, *.cpp, the Makefile and
    include/ocr_hip.h, in name order.  The counter summaries under profiles/ carry the fingerprint of the
    sources they were measured on (scripts/pmc_*.py, clock_diag.py); bench.py reports them as measurements
    only while it still matches, so a changed kernel cannot travel with stale counters.  (A content hash,
    not `git log`: the GPU box's copy of the repo has no .git.)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(_HERE, "csrc")
    files = sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h")) +
                   glob.glob(os.path.join(src, "*.cpp")) + [os.path.join(src, "Makefile")])
    files.append(os.path.join(os.path.dirname(_HERE), "include", "ocr_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
