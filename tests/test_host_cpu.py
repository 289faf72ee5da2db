"""CPU tests of the host logic: the C-ABI library loads and exports every symbol that
include/ocr_hip.h declares (no compute without a GPU), TF padding helper, checkpoint name
mapping, bucket planning, LR schedules, and the world_size-2 gloo all-reduce path."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="ocr_hip.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ocr_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from tensorflow_ocr_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 35
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.ocr_abi_version.restype = ctypes.c_int
    # the header's OCR_ABI_VERSION, the library's answer and the value the Python host checks in _lib.load() are one number
    hdr = int(re.search(r"#define OCR_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "ocr_hip.h")).read()).group(1))
    assert lib.ocr_abi_version() == hdr == _lib.ABI_VERSION
    lib.ocr_status_string.restype = ctypes.c_char_p
    assert lib.ocr_status_string(0) == b"ok" and b"unsupported" in lib.ocr_status_string(-2)
    lib.ocr_storage_dtype.restype = ctypes.c_char_p
    assert lib.ocr_storage_dtype() == b"f16"
    # the bfloat16 build of the same sources exports the same ABI
    bf = ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libocr_hip_bf16.so"))
    assert not [n for n in names if not hasattr(bf, n)]
    bf.ocr_storage_dtype.restype = ctypes.c_char_p
    assert bf.ocr_storage_dtype() == b"bf16"


def test_loader_refuses_a_library_of_another_abi_version(monkeypatch):
    """entry points changed their argument lists; a stale libocr_hip.so must be refused at load time, not
    called with its pointers shifted by a slot."""
    from tensorflow_ocr_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(_lib.OcrHipError, match="ABI version"):
        _lib.load()


def test_verification_kernels_live_in_their_own_library():
    """The plain direct f32 convolution (include/ocr_verify.h) is test infrastructure — the independent checker of the
    product library's matrix-core f32 precision: libocr_verify.so exports it, the product libraries do not; the f32
    precision itself (ocr_conv2d_f32_mfma + the element-wise ocr_*_f32 kernels) is declared in ocr_hip.h and exported by
    the product libraries."""
    from tensorflow_ocr_amd import _lib
    names = _declared_symbols("ocr_verify.h")
    assert names == ["ocr_conv2d_f32"]
    ver = ctypes.CDLL(_lib.VERIFY_LIB_PATH)
    assert not [n for n in names if not hasattr(ver, n)]
    prod_f32 = [n for n in _declared_symbols() if n.endswith("_f32_mfma") or n in (
        "ocr_channel_stats_f32", "ocr_bn_relu_f32", "ocr_maxpool_f32", "ocr_prep_images_f32", "ocr_bn_add_relu_f32", "ocr_unpool_f32")]
    assert len(prod_f32) == 7
    for path in (_lib.LIB_PATH, os.path.join(os.path.dirname(_lib.LIB_PATH), "libocr_hip_bf16.so")):
        prod = ctypes.CDLL(path)
        assert not [n for n in names if hasattr(prod, n)], path
        assert not [n for n in prod_f32 if not hasattr(prod, n)], path
    assert not set(names) & set(_declared_symbols())


def test_abi_rejects_bad_arguments_without_touching_the_gpu():
    from tensorflow_ocr_amd import _lib
    lib = _lib.load()
    lib.ocr_conv2d_f16.restype = ctypes.c_int
    d = _lib.ConvDesc(1, 8, 8, 24, 8, 8, 64, 3, 3, 1, 1, 1, 1, 0, 0)      # cin % 32 != 0
    assert lib.ocr_conv2d_f16(ctypes.byref(d), None, None, None, None, None, None) == -2
    d = _lib.ConvDesc(1, 8, 8, 64, 8, 8, 64, 3, 3, 1, 1, 1, 1, 0, 0)
    assert lib.ocr_conv2d_f16(ctypes.byref(d), None, None, None, None, None, None) == -1   # NULL x
    with pytest.raises(_lib.OcrHipError):
        _lib.call("ocr_conv2d_f16", ctypes.byref(d), None, None, None, None, None, None)


def test_product_path_has_no_oracle_import():
    pkg = os.path.join(ROOT, "tensorflow_ocr_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), os.path.join(dp, f)


def test_same_pad_matches_oracle():
    from oracle import ocr_oracle as O
    from tensorflow_ocr_amd import ops
    for size in (1, 7, 32, 33, 320, 512):
        for k, s, d in ((3, 1, 1), (2, 2, 1), (3, 2, 1), (3, 1, 6), (1, 2, 1), (7, 2, 1)):
            out, before, _ = O.tf_same_pad(size, k, s, d)
            assert ops.same_pad(size, k, s, d) == (out, before)


def test_checkpoint_name_mapping_roundtrip():
    from tensorflow_ocr_amd import checkpoint
    rng = np.random.default_rng(0)
    tf_sd = {"feature_fusion/Conv/weights": rng.standard_normal((1, 1, 8, 2)).astype(np.float32),
             "feature_fusion/Conv_5/weights": rng.standard_normal((1, 1, 8, 16)).astype(np.float32),
             "feature_fusion/Conv/BatchNorm/gamma": rng.standard_normal(2).astype(np.float32),
             "feature_fusion/Conv_5/BatchNorm/gamma": rng.standard_normal(16).astype(np.float32),
             "conv1/conv1_1/weights": rng.standard_normal((3, 3, 3, 4)).astype(np.float32)}
    internal = ["feature_fusion/Conv+Conv_5/weights", "feature_fusion/Conv+Conv_5/BatchNorm/gamma",
                "conv1/conv1_1/weights"]
    isd = checkpoint.tf_to_internal(internal, tf_sd)
    assert isd["feature_fusion/Conv+Conv_5/weights"].shape == (8, 18)
    assert np.array_equal(isd["feature_fusion/Conv+Conv_5/weights"][:, :2], tf_sd["feature_fusion/Conv/weights"][0, 0])
    back = checkpoint.internal_to_tf(isd)
    for k, v in tf_sd.items():
        assert np.array_equal(back[k], v), k
    # EAST's two sigmoid heads on the merge branch's output, merged into one variable (round 4): F_score 1 + geo_map 8
    east = {"feature_fusion/Conv_7/weights": rng.standard_normal((1, 1, 32, 1)).astype(np.float32),
            "feature_fusion/Conv_8/weights": rng.standard_normal((1, 1, 32, 8)).astype(np.float32),
            "feature_fusion/Conv_7/biases": rng.standard_normal(1).astype(np.float32),
            "feature_fusion/Conv_8/biases": rng.standard_normal(8).astype(np.float32)}
    isd = checkpoint.tf_to_internal(["feature_fusion/Conv_7+Conv_8/weights", "feature_fusion/Conv_7+Conv_8/biases"], east)
    assert isd["feature_fusion/Conv_7+Conv_8/weights"].shape == (32, 9) and isd["feature_fusion/Conv_7+Conv_8/biases"].shape == (9,)
    back = checkpoint.internal_to_tf(isd)
    for k, v in east.items():
        assert np.array_equal(back[k], v), k


def test_bucket_planner_covers_buffer_on_variable_boundaries():
    from tensorflow_ocr_amd.dist import plan_buckets
    ranges = [(0, 10), (10, 30), (30, 35), (35, 100)]
    for cap in (1, 20, 40, 1000):
        b = plan_buckets(ranges, 100, cap)
        assert b[0][0] == 0 and b[-1][1] == 100
        assert all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
        starts = {s for s, _ in ranges} | {100}
        assert all(s in starts and e in starts for s, e in b)
    assert plan_buckets(ranges, 100, 1000) == [(0, 100)]


def test_lr_schedules():
    from tensorflow_ocr_amd.train import exponential_decay, pixellink_lr
    assert exponential_decay(1e-4, 4999) == 1e-4
    assert abs(exponential_decay(1e-4, 10000) - 1e-4 * 0.94 ** 2) < 1e-15
    assert abs(pixellink_lr(0) - 1e-3) < 1e-12 and abs(pixellink_lr(20000) - 1e-4) < 1e-12
    assert abs(pixellink_lr(40000) - 1e-5) < 1e-12 and pixellink_lr(60000) == 0.01


_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as td
from tensorflow_ocr_amd import dist
from tensorflow_ocr_amd.graph import VariableStore, constant
rank, world, _ = dist.init_process_group_from_env("gloo")
st = VariableStore(torch.device("cpu"))
a = st.get("a/weights", (3, 3, 2, 4), constant(1.0), regularized=True)
b = st.get("b/weights", (5, 7), constant(2.0), regularized=True)
c = st.get("a/BatchNorm/gamma", (4,), constant(1.0))
st.materialise()
red = dist.GradientAllReduce(st, world, bucket_bytes=64 * 4, op="mean")
assert len(red.buckets) >= 2
for step in range(2):
    a.grad.fill_(float(rank + 1)); b.grad.fill_(10.0 * (rank + 1)); c.grad.fill_(100.0 * (rank + 1))
    red.on_grads_ready([b]); red.on_grads_ready([a, c])          # reverse creation order
    red.finish()
    mean = (1 + world) / 2.0
    assert torch.allclose(a.grad, torch.full_like(a.grad, mean)), a.grad
    assert torch.allclose(b.grad, torch.full_like(b.grad, 10 * mean))
    assert torch.allclose(c.grad, torch.full_like(c.grad, 100 * mean))
# fold_mean leaves the SUM and reports the scale
red2 = dist.GradientAllReduce(st, world, fold_mean=True)
a.grad.fill_(float(rank + 1)); red2.finish()
assert torch.allclose(a.grad, torch.full_like(a.grad, world * (world + 1) / 2.0)) and red2.grad_scale == 1.0 / world
td.barrier(); td.destroy_process_group()
print("rank", rank, "ok")
"""


def test_gradient_allreduce_world2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER % ROOT)
    import socket

    def launch():
        with socket.socket() as sk:                  # a port nobody holds right now
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        ps = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            ps.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                       stderr=subprocess.STDOUT))
        try:
            return ps, [p.communicate(timeout=120)[0].decode() for p in ps]
        except subprocess.TimeoutExpired:
            for p in ps:
                p.kill()
            return ps, None
    procs, outs = launch()
    if outs is None:
        procs, outs = launch()                       # one more try on a fresh port
    assert outs is not None, "two-rank rendezvous timed out twice"
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "ok" in o


# ------------------------------------------------------------------ self-launch (one process per GPU)
_LAUNCH_SCRIPT = r"""
import json, os, sys
sys.path.insert(0, %r)
from tensorflow_ocr_amd import launch
n = int(sys.argv[1]); mode = sys.argv[2]
rc = launch.self_launch(n)            # launcher: returns the job's exit code; rank: returns None
if rc is not None:
    sys.exit(rc)
import torch, torch.distributed as td
from tensorflow_ocr_amd import dist
rank, world, local = dist.init_process_group_from_env("gloo")
assert world == n and local == rank and os.environ["MASTER_ADDR"] == "127.0.0.1"
if mode == "fail" and rank == 1:
    sys.exit(7)                       # one tower dies before the collective: the others must not hang
t = torch.ones(1); td.all_reduce(t)
stop = dist.any_rank(rank == world - 1)            # only the LAST rank saw a NaN: everybody stops
none = dist.any_rank(False)
print("noise from rank", rank) if rank else None   # non-zero ranks' stdout goes to stderr
if rank == 0:
    print(json.dumps({"ranks": int(t.item()), "stop": stop, "none": none}))
td.barrier(); td.destroy_process_group()
"""


def _run_launcher(tmp_path, n, mode, timeout=240):
    script = tmp_path / "selflaunch.py"
    script.write_text(_LAUNCH_SCRIPT % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    return subprocess.run([sys.executable, str(script), str(n), mode], env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=timeout)


def test_self_launch_spawns_ranks_and_relays_rank0(tmp_path):
    """`python bench.py --gpus N` from a clean shell: the parent spawns N ranks (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their env), only rank 0's stdout reaches the parent's stdout, exit 0."""
    import json
    r = _run_launcher(tmp_path, 2, "ok")
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    # (gloo itself prints a "[Gloo] Rank 0 is connected ..." banner on stdout; RCCL does not)
    lines = [l for l in r.stdout.decode().splitlines() if l.strip() and not l.startswith("[Gloo]")]
    assert len(lines) == 1, lines                       # the ONE JSON line
    out = json.loads(lines[0])
    assert out == {"ranks": 2, "stop": True, "none": False}
    assert b"noise from rank 1" in r.stderr


def test_self_launch_fails_when_a_rank_fails(tmp_path):
    r = _run_launcher(tmp_path, 2, "fail", timeout=120)
    assert r.returncode == 7
    assert b"rank 1 exited with code 7" in r.stderr and b"{" not in r.stdout


def test_self_launch_is_a_noop_for_ranks_and_single_tower(monkeypatch):
    from tensorflow_ocr_amd import launch
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert launch.self_launch(1) is None
    monkeypatch.setenv("WORLD_SIZE", "4")
    assert launch.self_launch(4) is None
    env = launch.child_env(3, 8, 1234, visible="0,1,2,3,4,5,6,7", base={"CUDA_VISIBLE_DEVICES": "5"})
    assert env["RANK"] == "3" and env["LOCAL_RANK"] == "3" and env["WORLD_SIZE"] == "8"
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "1234"
    assert env["HIP_VISIBLE_DEVICES"] == "0,1,2,3,4,5,6,7" and "CUDA_VISIBLE_DEVICES" not in env
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


_SUM_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as td
from tensorflow_ocr_amd import dist
from tensorflow_ocr_amd.graph import VariableStore, constant
rank, world, _ = dist.init_process_group_from_env("gloo")
st = VariableStore(torch.device("cpu"))
a = st.get("a/weights", (3, 3, 2, 4), constant(1.0), regularized=True)
b = st.get("b/biases", (6,), constant(0.0))
st.materialise()
# train_pixellink.py:179-194,264: every clone differentiates loss / num_clones, the gradients are SUMMED
red = dist.GradientAllReduce(st, world, bucket_bytes=64 * 4, op="sum", fold_mean=True)
assert red.grad_scale == 1.0 and red.active
a.grad.fill_((rank + 1) / world); b.grad.fill_(10.0 * (rank + 1) / world)
red.on_grads_ready([b]); red.on_grads_ready([a]); red.finish()
mean = (1 + world) / 2.0
assert torch.allclose(a.grad, torch.full_like(a.grad, mean)) and torch.allclose(b.grad, torch.full_like(b.grad, 10 * mean))
# disabled (bench.py's comm-exposed A/B): hooks are no-ops, the local gradient stays
red.enabled = False
a.grad.fill_(float(rank)); red.on_grads_ready([a, b]); red.finish()
assert torch.allclose(a.grad, torch.full_like(a.grad, float(rank)))
assert sum(red.bucket_nbytes()) == st.flat.numel() * 4
td.barrier(); td.destroy_process_group()
print("rank", rank, "ok")
"""


def test_gradient_sum_semantics_world2_gloo(tmp_path):
    script = tmp_path / "s.py"
    script.write_text(_SUM_WORKER % ROOT)
    from tensorflow_ocr_amd import launch
    port = launch.free_port()
    ps = [subprocess.Popen([sys.executable, str(script)], env=launch.child_env(r, 2, port), stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT) for r in range(2)]
    outs = []
    for p in ps:
        try:
            outs.append(p.communicate(timeout=120)[0].decode())
        except subprocess.TimeoutExpired:
            for q in ps:
                q.kill()
            raise
    for p, o in zip(ps, outs):
        assert p.returncode == 0 and "ok" in o, o[-2000:]


def test_one_rank_group_can_be_forced(tmp_path):
    """OCR_FORCE_PG / force=True: a ONE-rank group (what the 1-GPU RCCL test uses) and a reducer that
    runs its bucket path at world 1."""
    code = r"""
import sys; sys.path.insert(0, %r)
import torch, torch.distributed as td
from tensorflow_ocr_amd import dist
from tensorflow_ocr_amd.graph import VariableStore, constant
assert dist.init_process_group_from_env("gloo") == (0, 1, 0) and not td.is_initialized()
rank, world, local = dist.init_process_group_from_env("gloo", force=True)
assert (rank, world, local) == (0, 1, 0) and td.is_initialized() and td.get_world_size() == 1
st = VariableStore(torch.device("cpu"))
a = st.get("a/weights", (8,), constant(1.0)); st.materialise()
red = dist.GradientAllReduce(st, 1, fold_mean=True, force=True)
assert red.active
a.grad.fill_(3.0); red.on_grads_ready([a]); assert red.fired == [True]; red.finish()
assert torch.allclose(a.grad, torch.full_like(a.grad, 3.0))
assert not dist.GradientAllReduce(st, 1).active
td.destroy_process_group(); print("ok")
""" % ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "OCR_FORCE_PG")}
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert r.returncode == 0 and b"ok" in r.stdout, r.stdout.decode()[-2000:]


def test_recorder_only_records_its_owner_thread():
    """a feeder-thread C-ABI call must never land in the training plan."""
    import threading
    from tensorflow_ocr_amd import _lib
    rec = _lib.Recorder()
    assert rec.mine()
    seen = []
    t = threading.Thread(target=lambda: (seen.append(rec.mine()), rec.py(lambda: None)))
    t.start(); t.join()
    assert seen == [False] and rec.entries == []
    rec.py(lambda: None)
    assert len(rec.entries) == 1


# ------------------------------------------------------------------ world 8 (the node the SCALE run uses)
def test_self_launch_world8_and_rank_failure_teardown(tmp_path):
    """The launcher at the driver's scale, on CPU: 8 children rendezvous over gloo, rank 0's single JSON
    line says ranks == 8; with one rank dying before the collective the other 7 are reaped and the
    launcher exits with that rank's code (no rank is left waiting in an all-reduce)."""
    import json
    r = _run_launcher(tmp_path, 8, "ok", timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip() and not l.startswith("[Gloo]")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"ranks": 8, "stop": True, "none": False}, lines
    for k in range(1, 8):
        assert ("noise from rank %d" % k).encode() in r.stderr
    r = _run_launcher(tmp_path, 8, "fail", timeout=120)
    assert r.returncode == 7 and b"rank 1 exited with code 7" in r.stderr and b"{" not in r.stdout


_HANG_SCRIPT = r"""
import os, sys, time
sys.path.insert(0, %r)
from tensorflow_ocr_amd import launch
rc = launch.self_launch(3)
if rc is not None:
    sys.exit(rc)
open(os.path.join(%r, "pid_%%s" %% os.environ["RANK"]), "w").write(str(os.getpid()))
time.sleep(600)                 # a rank stuck in a collective
"""


def test_launcher_sigterm_reaps_its_ranks(tmp_path):
    """a cancelled job (SIGTERM to the launcher) must not orphan the ranks."""
    import signal
    import time
    script = tmp_path / "hang.py"
    script.write_text(_HANG_SCRIPT % (ROOT, str(tmp_path)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.Popen([sys.executable, str(script)], env=env, stderr=subprocess.PIPE)
    t0 = time.time()
    while len([f for f in os.listdir(tmp_path) if f.startswith("pid_")]) < 3 and time.time() - t0 < 60:
        time.sleep(0.1)
    pids = [int(open(os.path.join(tmp_path, f)).read()) for f in os.listdir(tmp_path) if f.startswith("pid_")]
    assert len(pids) == 3
    p.send_signal(signal.SIGTERM)
    err = p.communicate(timeout=60)[1]
    assert p.returncode == 128 + signal.SIGTERM, (p.returncode, err.decode()[-1000:])
    assert b"stopping 3 ranks" in err
    time.sleep(0.2)
    for pid in pids:
        alive = True
        try:
            os.kill(pid, 0)
            # a zombie of another parent cannot exist here: the launcher waited for its children
        except ProcessLookupError:
            alive = False
        assert not alive, pid


def test_child_env_keeps_the_scheduler_restriction():
    """--gpu_list narrows HIP_VISIBLE_DEVICES inside ROCR_VISIBLE_DEVICES, it never lifts it."""
    from tensorflow_ocr_amd import launch
    env = launch.child_env(1, 2, 999, visible="0,1", base={"ROCR_VISIBLE_DEVICES": "4,5", "CUDA_VISIBLE_DEVICES": "3"})
    assert env["ROCR_VISIBLE_DEVICES"] == "4,5" and env["HIP_VISIBLE_DEVICES"] == "0,1" and "CUDA_VISIBLE_DEVICES" not in env


def test_exchange_abi_argument_checks_without_a_gpu():
    """ocr_allreduce_bucket / ocr_comm_* / ocr_event_* reject bad arguments before touching RCCL or HIP."""
    from tensorflow_ocr_amd import _lib
    lib = _lib.load()
    for n in ("ocr_allreduce_bucket", "ocr_comm_init_rank", "ocr_comm_unique_id", "ocr_event_record",
              "ocr_stream_wait_event", "ocr_comm_destroy", "ocr_event_destroy"):
        getattr(lib, n).restype = ctypes.c_int
    assert lib.ocr_allreduce_bucket(None, None, ctypes.c_size_t(4), 0, 0, None) == -1          # NULL communicator
    fake = ctypes.c_void_p(1)
    assert lib.ocr_allreduce_bucket(fake, None, ctypes.c_size_t(4), 0, 0, None) == -1          # NULL buffer
    buf = ctypes.c_void_p(4096)
    assert lib.ocr_allreduce_bucket(fake, buf, ctypes.c_size_t(4), 99, 0, None) == -1          # unknown dtype
    assert lib.ocr_allreduce_bucket(fake, buf, ctypes.c_size_t(4), 0, 99, None) == -1          # unknown op
    assert lib.ocr_allreduce_bucket(fake, buf, ctypes.c_size_t(0), 0, 0, None) == 0            # empty bucket
    h = ctypes.c_void_p()
    ident = ctypes.create_string_buffer(128)
    assert lib.ocr_comm_init_rank(ctypes.byref(h), 0, ident, 0) == -1                           # nranks < 1
    assert lib.ocr_comm_init_rank(ctypes.byref(h), 2, ident, 2) == -1                           # rank out of range
    assert lib.ocr_comm_unique_id(None) == -1
    assert lib.ocr_event_record(None, None) == -1 and lib.ocr_stream_wait_event(None, None) == -1
    assert lib.ocr_comm_destroy(None) == 0 and lib.ocr_event_destroy(None) == 0
    lib.ocr_status_string.restype = ctypes.c_char_p
    assert b"RCCL" in lib.ocr_status_string(-5)


def test_exchange_mode_selection(monkeypatch):
    from tensorflow_ocr_amd import dist
    monkeypatch.delenv("OCR_EXCHANGE", raising=False)
    assert dist.exchange_mode(8, cuda=False) == "torch"           # gloo / CPU towers
    assert dist.exchange_mode(1, cuda=True) == "torch"            # a single tower exchanges nothing
    assert dist.exchange_mode(1, cuda=True, force=True) == "abi"  # one-rank RCCL communicator through the C ABI
    assert dist.exchange_mode(8, cuda=True) == "abi"
    monkeypatch.setenv("OCR_EXCHANGE", "torch")
    assert dist.exchange_mode(8, cuda=True) == "torch"


_STORE_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch.distributed as td
from tensorflow_ocr_amd import dist
rank, world, _ = dist.init_process_group_from_env("gloo")
# the 128-byte RCCL unique id is arbitrary binary (NULs, bytes > 0x7f): the rendezvous dist.AbiComm uses must carry it intact
ident = np.random.default_rng(5).integers(0, 256, 128).astype(np.uint8).tobytes()
assert b"\x00" in ident or True
store = td.distributed_c10d._get_default_store()
if rank == 0:
    store.set("ocr_comm_id_test", ident)
got = bytes(store.get("ocr_comm_id_test"))
assert got == ident and len(got) == 128, (len(got), got[:8], ident[:8])
import ctypes
buf = ctypes.create_string_buffer(got, 128)
assert buf.raw == ident
td.barrier(); td.destroy_process_group()
print("rank", rank, "ok")
"""


def test_unique_id_rendezvous_is_binary_safe(tmp_path):
    """dist.AbiComm hands rank 0's ocr_comm_unique_id bytes to the other ranks through torch.distributed's store:
    128 arbitrary bytes must arrive unchanged (world 2, gloo, CPU)."""
    script = tmp_path / "store.py"
    script.write_text(_STORE_WORKER % ROOT)
    from tensorflow_ocr_amd import launch
    port = launch.free_port()
    ps = [subprocess.Popen([sys.executable, str(script)], env=launch.child_env(r, 2, port), stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT) for r in range(2)]
    outs = []
    for p in ps:
        try:
            outs.append(p.communicate(timeout=120)[0].decode())
        except subprocess.TimeoutExpired:
            for q in ps:
                q.kill()
            raise
    for p, o in zip(ps, outs):
        assert p.returncode == 0 and "ok" in o, o[-2000:]


def test_steady_stats_script_drops_the_build_steps(tmp_path):
    """scripts/steady_stats.py: per-kernel statistics from a rocprofv3 kernel trace AFTER the first N steps (a step ends
    at the optimiser launch), same columns as rocprofv3's own table — the cold first launch of a kernel (lazy code-object
    load) must not reach the average the roofline is checked against."""
    import csv
    d = tmp_path / "prof" / "host"
    d.mkdir(parents=True)
    rows, t = [], 1000
    for step in range(5):
        for name, dur in (("conv3x3_w4_kernel(...)", 30000000 if step == 0 else 350000), ("bn_relu_kernel<1,0>", 90000),
                          ("adam_kernel(AdamP)", 120000)):
            rows.append({"Kernel_Name": name, "Start_Timestamp": t, "End_Timestamp": t + dur, "Dispatch_Id": len(rows) + 1})
            t += dur + 1000
    with open(d / "1_kernel_trace.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "steady_stats.py"), str(tmp_path / "prof"),
                        "--skip-steps", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 0, r.stderr.decode()
    out = {x["Name"]: x for x in csv.DictReader(r.stdout.decode().splitlines())}
    assert set(out) == {"conv3x3_w4_kernel(...)", "bn_relu_kernel<1,0>", "adam_kernel(AdamP)"}
    assert int(out["conv3x3_w4_kernel(...)"]["Calls"]) == 3 and float(out["conv3x3_w4_kernel(...)"]["AverageNs"]) == 350000.0
    assert b"3 steps" in r.stderr
    assert abs(sum(float(x["Percentage"]) for x in out.values()) - 100.0) < 1e-3


def test_counter_provenance_gates_the_reported_fields(tmp_path, monkeypatch):
    """bench.counters_from_profiles: a profiler-derived field is reported only while the file's `_provenance.csrc_sha16`
    equals the fingerprint of the current kernel sources."""
    import importlib
    import json
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    from tensorflow_ocr_amd import _lib
    now = _lib.csrc_fingerprint()
    assert len(now) == 16 and now == _lib.csrc_fingerprint()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    dom = "conv3x3_w4_kernel"
    good = {"_provenance": {"csrc_sha16": now, "date": "d"}, dom: {"hbm_bytes_per_launch": 123, "mfma_busy_frac": 0.5}}
    (prof / (bench.PROFILE_ROUND + "_pmc_traffic.json")).write_text(json.dumps(good))
    stale = {"_provenance": {"csrc_sha16": "0" * 16, "date": "d"}, dom: {"mfma_busy_frac": 0.9}}
    (prof / (bench.PROFILE_ROUND + "_pmc_mfma.json")).write_text(json.dumps(stale))
    vals, src = bench.counters_from_profiles(dom)
    assert vals == {"traffic": 123, "mfma_busy": None, "clock_ghz": None}
    assert src["files"]["traffic"]["current"] and not src["files"]["mfma_busy"]["current"]
    assert src["files"]["clock_ghz"]["file"] is None and src["csrc_sha16_now"] == now


def test_design_tables_show_the_committed_profiles(tmp_path):
    """DESIGN.md section 5's per-layer table of the dominant kernel and its per-step kernel table are GENERATED from
    this round's profiles/rNN_* (scripts/design_tables.py; NN = bench.PROFILE_ROUND): regenerating them changes nothing — the text cannot drift from the
    committed measurements — and the kernel-name folding puts the epilogue-mode instantiations under one name."""
    import importlib.util
    import shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("design_tables", os.path.join(root, "scripts", "design_tables.py"))
    dt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dt)
    sys.path.insert(0, root)
    import bench
    rnd = bench.PROFILE_ROUND
    before = open(os.path.join(root, "DESIGN.md")).read()
    for name, text in (("w4_per_layer", dt.w4_table(rnd)), ("step_kernels", dt.step_table(rnd))):
        start = before.index("<!-- generated: %s -->\n" % name) + len("<!-- generated: %s -->\n" % name)
        end = before.index("\n<!-- end generated -->", start)
        assert before[start:end] == text, name
    assert "| `conv3x3_w4_kernel` | 17 |" in dt.step_table(rnd)
    sys.path.insert(0, os.path.join(root, "scripts"))
    from pmc_mfma import short
    assert short("_ZN12_GLOBAL__N_117conv3x3_w4_kernelILi2EEEvNS_5ConvPEPKDF16_S3_PKfPDF16_Pf") == "conv3x3_w4_kernel"
    assert short("void conv3x3_w4s_kernel<128, 1>(ConvP, bool)") == "conv3x3_w4s_kernel<128>"
    assert short("void conv_c64_persist_kernel<64, false, 6>(ConvP)") == "conv_c64_persist_kernel<64>"
    assert short("void conv_pw_kernel<256, 4, true>(ConvP)") == "conv_pw_kernel<256,4,1>"


def test_guest_kernels_fit_beside_the_weight_gradient():
    """csrc/guest_bn.hip's kernels are placed beside a resident wgrad3_kernel<9,128> workgroup (198 VGPR + 256 AGPR ->
    200 + 256 = 456 of the 512 registers per lane a SIMD has; allocation granule 8): they must stay within 56 registers
    per lane, a few bytes of LDS (the work-queue index) and no scratch, in both storage builds — and the host kernel must not have grown."""
    csrc = os.path.join(ROOT, "tensorflow_ocr_amd", "csrc")

    def usage(src, extra=()):
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-DOCR_WPS=1", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                            "-Rpass-analysis=kernel-resource-usage", *extra, "-c", os.path.join(csrc, src), "-o", os.devnull],
                           capture_output=True, text=True, cwd=csrc)
        assert r.returncode == 0, r.stderr[-2000:]
        out = {}
        for blk in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
            name = blk.split()[0]
            f = lambda k: int(re.search(k + r": (\d+)", blk).group(1))
            out[name] = dict(vgpr=f("VGPRs"), agpr=f("AGPRs"), scratch=f(r"ScratchSize \[bytes/lane\]"), lds=f(r"LDS Size \[bytes/block\]"))
        return out
    for extra in ((), ("-DOCR_BF16",)):
        guests = {k: v for k, v in usage("guest_bn.hip", extra).items() if "affine" in k}
        assert len(guests) == 5, guests.keys()
        for k, v in guests.items():
            alloc = (v["vgpr"] + 7) // 8 * 8
            assert alloc <= 56 and v["agpr"] == 0 and v["scratch"] == 0 and v["lds"] <= 64, (k, v)
    host = {k: v for k, v in usage("conv_wgrad.hip").items() if "wgrad3_kernelILi9ELi128" in k}
    assert len(host) == 1
    v = next(iter(host.values()))
    assert (v["vgpr"] + 3) // 4 * 4 + v["agpr"] <= 456 and v["scratch"] == 0, v


def test_schedule_guests_holds_weight_gradients_back_and_spends_them_as_hosts():
    """train.schedule_guests on a synthetic recorded plan: weight gradients wait (with what was recorded behind them);
    every guest forks, goes to the second stream and takes held-back slab kernels along — oldest first, as many as cover
    it without a large overshoot; the join sits in front of the next main-stream call and the slab sums / exchange
    entries of the hosts follow it; whatever is still held back is issued in front of the first entry that needs the
    gradients; a guest with nothing to run beside stays serial."""
    from tensorflow_ocr_amd.train import schedule_guests

    def c(name, tag=None):
        return ["c", None, (), name, tag]
    W = 1.3e9 * 100            # a weight gradient estimated at 100 us
    G = 5.0e6 * 100            # a guest that takes 100 us alone
    plan = [c("apply5", ("guest", G)),                                       # nothing held back yet: serial
            c("dgrad5"), c("wgrad5", ("side", W)), c("red5", ("reduce",)), c("s2d5", ("side",)), c("x5a", ("xchg", None, "early")),
            c("x5b", ("xchg", "rccl", "early")),
            c("coef4", ("pre",)), c("apply4", ("guest", G)), c("dgrad4"), c("wgrad4", ("side", W)), c("red4", ("reduce",)),
            c("coef3", ("pre",)), c("apply3", ("guest", 0.25 * G)), c("dgrad3"), c("wgrad3", ("side", W)), c("wgrad3b", ("side", 5 * W)),
            c("dgrad2"), c("wgrad2", ("side", W)),
            c("xl", ("xchg", "rccl", "late")), c("xf", ("xchg", "finish")), ["py", None, "opt"]]
    out = schedule_guests(plan, cover=2.0, min_us=0, xchg_at_fork=False)
    names = [e[3] if e[0] == "c" else e[0] for e in out]
    assert names == ["apply5", "dgrad5",
                     "coef4", "fork", "apply4", "wgrad5", "join", "red5", "s2d5", "x5a", "x5b",   # one host is all there is
                     "dgrad4", "coef3", "fork", "apply3", "wgrad4", "join", "red4",               # 25 us x 2: the smallest host
                     "dgrad3", "wgrad3", "wgrad3b", "dgrad2", "wgrad2", "xl", "xf", "py"]         # no guest ahead: in place
    paired = [e[3] for e in out if e[0] == "c" and e[4] is not None and e[4][0] == "guest" and e[4][-1] == "paired"]
    assert paired == ["apply4", "apply3"]
    assert plan[8][4] == ("guest", G)                                        # the input plan is not modified
    # a long guest takes several hosts along, skips one that would overshoot, and leaves the rest held back
    plan2 = [c("w0", ("side", W)), c("big", ("side", 4 * W)), c("w1", ("side", W)), c("w2", ("side", W)), c("w3", ("side", W)),
             c("coef", ("pre",)), c("apply", ("guest", 1.4 * G)), c("dgrad"), ["py", None]]
    names2 = [e[3] if e[0] == "c" else e[0] for e in schedule_guests(plan2, cover=2.0, min_us=0, xchg_at_fork=False)]
    assert names2 == ["coef", "fork", "apply", "w0", "w1", "w2", "join", "dgrad", "big", "w3", "py"]
    # a host far larger than the guest needs is not spent on it
    plan3 = [c("huge", ("side", 10 * W)), c("coef", ("pre",)), c("apply", ("guest", 0.2 * G)), c("dgrad"), ["py", None]]
    assert [e[3] if e[0] == "c" else e[0] for e in schedule_guests(plan3, cover=2.0, min_us=0, xchg_at_fork=False)] == ["coef", "apply", "dgrad", "huge", "py"]
    # a host whose weight gradient completed an exchange bucket never overtakes an older held-back weight gradient (the
    # bucket's all-reduce reads both): the small one stays behind the big one it was recorded after
    plan6 = [c("big", ("side", 4 * W)), c("small", ("side", W)), c("xs", ("xchg", "rccl", "early")), c("coef", ("pre",)),
             c("apply", ("guest", 0.5 * G)), c("dgrad"), ["py", None]]
    assert [e[3] if e[0] == "c" else e[0] for e in schedule_guests(plan6, cover=2.0, min_us=0, xchg_at_fork=False)] == [
        "coef", "apply", "dgrad", "big", "small", "xs", "py"]
    plan7 = [c("big", ("side", 4 * W)), c("small", ("side", W)), c("coef", ("pre",)), c("apply", ("guest", 0.5 * G)), c("dgrad"), ["py", None]]
    assert [e[3] if e[0] == "c" else e[0] for e in schedule_guests(plan7, cover=2.0, min_us=0, xchg_at_fork=False)] == [
        "coef", "fork", "apply", "small", "join", "dgrad", "big", "py"]
    # no guest ahead: nothing is held back (a net without batch norm keeps its recorded order, exchange entries included)
    plan5 = [c("dgrad"), c("w0", ("side", W)), c("r0", ("reduce",)), c("x0", ("xchg", None, "early")), c("dgrad1"), c("w1", ("side", W)),
             c("xf", ("xchg", "finish")), ["py", None]]
    assert [e[3] if e[0] == "c" else e[0] for e in schedule_guests(plan5, cover=2.0, min_us=0, xchg_at_fork=False)] == [
        "dgrad", "w0", "r0", "x0", "dgrad1", "w1", "xf", "py"]
    # a guest shorter than a fork + join costs is left alone; a weight gradient that cannot host stays where it was recorded
    plan4 = [c("w0", ("side", W)), c("fat", ("side",)), c("redf", ("reduce",)), c("coef", ("pre",)), c("apply", ("guest", 0.3 * G)),
             c("dgrad"), ["py", None]]
    assert [e[3] if e[0] == "c" else e[0] for e in schedule_guests(plan4, cover=2.0, min_us=40, xchg_at_fork=False)] == [
        "coef", "apply", "dgrad", "w0", "fat", "redf", "py"]


def test_schedule_guests_never_fires_a_bucket_before_its_held_back_weight_gradients():
    """an exchange entry that travels behind a weight gradient which stays IN PLACE (one that cannot
    host: ("side",) without a FLOP count, an untagged one such as conv1_1's sums form, or one with no guest ahead) used to
    be issued while older weight gradients of the same bucket were still held back, so the bucket's all-reduce read
    gradients that had not been written.  Property checked on several plans: in the scheduled order every weight-gradient
    entry recorded BEFORE an exchange entry is still in front of it, and the multiset of entries is unchanged."""
    from tensorflow_ocr_amd.train import schedule_guests

    def c(name, tag=None):
        return ["c", None, (), name, tag]
    W, G = 1.3e9 * 100, 5.0e6 * 100

    def check(plan, **kw):
        check1(plan, xchg_at_fork=True, **kw)
        return check1(plan, xchg_at_fork=False, **kw)

    def check1(plan, **kw):
        out = schedule_guests(plan, **kw)
        names = [e[3] if e[0] == "c" else e[0] for e in out]
        rec = [e[3] if e[0] == "c" else e[0] for e in plan]
        assert sorted(n for n in names if n not in ("fork", "join")) == sorted(rec)
        for xi, e in enumerate(plan):
            if e[0] == "c" and e[4] is not None and e[4][0] == "xchg":
                for w in plan[:xi]:
                    if w[0] == "c" and (w[3].startswith("w") or w[3].startswith("red")):
                        assert names.index(w[3]) < names.index(e[3]), (w[3], e[3], names)
        return names
    # the advisor's plan: wA is held back (a small guest follows that takes nothing), wB cannot host and stays in place
    # with the bucket's early all-reduce behind it
    plan = [c("wA", ("side", W)), c("redA", ("reduce",)), c("coef", ("pre",)), c("apply", ("guest", 0.1 * G)), c("dgrad"),
            c("wB", ("side",)), c("redB", ("reduce",)), c("xAB", ("xchg", "rccl", "early")), c("xf", ("xchg", "finish")), ["py", None, "opt"]]
    names = check(plan, cover=2.0, min_us=40)
    assert names == ["coef", "apply", "dgrad", "wB", "redB", "wA", "redA", "xAB", "xf", "py"]
    # every guest below the minimum: all weight gradients are held back until something needs them — here the bucket that
    # conv1_1's untagged sums-form weight gradient completes
    plan = [c("w3", ("side", W)), c("red3", ("reduce",)), c("coef2", ("pre",)), c("apply2", ("guest", 0.1 * G)), c("dgrad2"),
            c("w2", ("side", W)), c("red2", ("reduce",)), c("coef1", ("pre",)), c("apply1", ("guest", 0.1 * G)),
            c("w1sums"), c("x0a", ("xchg", None, "early")), c("x0b", ("xchg", "rccl", "early")), c("xf", ("xchg", "finish")), ["py", None, "opt"]]
    names = check(plan, cover=2.0, min_us=40)
    assert names.index("w3") < names.index("x0a") and names.index("w2") < names.index("x0a")
    # a weight gradient with no guest ahead stays in place; the early entries behind it still wait for the held-back ones
    plan = [c("w5", ("side", W)), c("red5", ("reduce",)), c("coef4", ("pre",)), c("apply4", ("guest", 0.1 * G)), c("dgrad4"),
            c("w4", ("side", W)), c("red4", ("reduce",)), c("x45", ("xchg", "rccl", "early")), c("dgrad3"), c("xf", ("xchg", "finish")), ["py", None]]
    check(plan, cover=2.0, min_us=40)
    # pairing still happens where it is safe, and a paired host's own exchange entries follow the join
    plan = [c("w5", ("side", W)), c("red5", ("reduce",)), c("x5", ("xchg", "rccl", "early")), c("coef4", ("pre",)),
            c("apply4", ("guest", G)), c("dgrad4"), c("w4", ("side",)), c("red4", ("reduce",)), c("x4", ("xchg", "rccl", "early")),
            c("xf", ("xchg", "finish")), ["py", None]]
    names = check(plan, cover=1.0, min_us=0)
    assert names[:6] == ["coef4", "fork", "apply4", "w5", "join", "red5"]


def test_schedule_guests_shares_hosts_out_when_the_guests_to_come_want_more_than_there_is():
    """Round 6 (train.GUEST_BALANCE): the greedy choice gave conv2_1's apply pass two weight gradients (260 + 525 us for a
    pass that wants 676) and left the last, largest pass 306 us for the 1 071 it wants.  With the guests still to come in
    view each gets its share: the first takes the ONE host closest to its share, the last gets the other two.  With hosts
    to spare the choice covers the guest (no deficit) with the smallest excess."""
    from tensorflow_ocr_amd.train import schedule_guests

    def c(name, tag=None):
        return ["c", None, (), name, tag]
    us = lambda t: 1.3e9 * t                 # a weight gradient estimated at t us
    al = lambda t: 5.0e6 * t                 # a guest that takes t us alone
    plan = [c("wA", ("side", us(260))), c("wB", ("side", us(525))),
            c("coef9", ("pre",)), c("g9", ("guest", al(322))), c("dgrad9"),
            c("wC", ("side", us(306))),
            c("coef10", ("pre",)), c("g10", ("guest", al(510))), c("dgrad10"),
            c("wD", ("side", us(537))), ["py", None, "opt"]]
    names = lambda p: [e[3] if e[0] == "c" else e[0] for e in p]
    greedy = names(schedule_guests(plan, cover=2.1, min_us=40, balance=False))
    assert greedy == ["coef9", "fork", "g9", "wA", "wB", "join", "dgrad9", "coef10", "fork", "g10", "wC", "join", "dgrad10", "wD", "py"]
    shared = names(schedule_guests(plan, cover=2.1, min_us=40, balance=True))
    assert shared == ["coef9", "fork", "g9", "wB", "join", "dgrad9", "coef10", "fork", "g10", "wA", "wC", "join", "dgrad10", "wD", "py"]
    # hosts to spare: cover, with the smallest excess (the 700 us host stays for nobody in particular)
    plan2 = [c("w1", ("side", us(700))), c("w2", ("side", us(150))), c("w3", ("side", us(120))),
             c("coef", ("pre",)), c("g", ("guest", al(100))), c("dgrad"), ["py", None, "opt"]]
    assert names(schedule_guests(plan2, cover=2.1, min_us=40, balance=True)) == [
        "coef", "fork", "g", "w2", "w3", "join", "dgrad", "w1", "py"]


def test_schedule_guests_exchange_at_the_next_fork():
    """xchg_at_fork: a paired host's exchange entries wait for the NEXT fork (they start beside that fork's hosts) or for
    the first entry that needs the gradients; nothing is lost, weight gradients still precede their bucket's exchange."""
    from tensorflow_ocr_amd.train import schedule_guests

    def c(name, tag=None):
        return ["c", None, (), name, tag]
    W, G = 1.3e9 * 100, 5.0e6 * 100
    plan = [c("w5", ("side", W)), c("red5", ("reduce",)), c("x5", ("xchg", "rccl", "early")),
            c("coef4", ("pre",)), c("apply4", ("guest", G)), c("dgrad4"),
            c("w4", ("side", W)), c("red4", ("reduce",)), c("x4", ("xchg", "rccl", "early")),
            c("coef3", ("pre",)), c("apply3", ("guest", G)), c("dgrad3"),
            c("w3", ("side", W)), c("red3", ("reduce",)),
            c("xf", ("xchg", "finish")), ["py", None, "opt"]]
    names = lambda p: [e[3] if e[0] == "c" else e[0] for e in p]
    base = names(schedule_guests(plan, cover=1.0, min_us=0, xchg_at_fork=False))
    assert base == ["coef4", "fork", "apply4", "w5", "join", "red5", "x5", "dgrad4",
                    "coef3", "fork", "apply3", "w4", "join", "red4", "x4", "dgrad3", "w3", "red3", "xf", "py"]
    at_fork = names(schedule_guests(plan, cover=1.0, min_us=0, xchg_at_fork=True))
    assert at_fork == ["coef4", "fork", "apply4", "w5", "join", "red5", "dgrad4",
                       "coef3", "fork", "x5", "apply3", "w4", "join", "red4", "dgrad3", "x4", "w3", "red3", "xf", "py"]      # (x4: no fork left)
    assert sorted(at_fork) == sorted(base)


def test_decode_slab_is_reserved_not_sparse(tmp_path):
    """the decode workers' slab must be RESERVED when it is created (a sparse file under a 64 MB /dev/shm maps
    fine and kills the process with SIGBUS at the first page past the limit).  _reserve_slab: pages allocated, fewer
    slots when the first size does not fit, the next directory when nothing fits, SlabUnavailable naming what was tried;
    a pool of decode processes round-trips an image through the reserved slab."""
    from tensorflow_ocr_amd.datasets import _decode
    slots, path = _decode._reserve_slab(8, 3, 1 << 16, "slab_a", dirs=(str(tmp_path),))
    try:
        st = os.stat(path)
        assert slots == 8 and st.st_size == 8 << 16 and st.st_blocks * 512 >= st.st_size      # allocated, not sparse
    finally:
        os.unlink(path)
    with pytest.raises(_decode.SlabUnavailable, match="missing or not writable"):
        _decode._reserve_slab(4, 2, 1 << 16, "slab_b", dirs=(str(tmp_path / "nope"),))
    # a directory that cannot hold the request: RLIMIT_FSIZE makes posix_fallocate fail with EFBIG in a child process
    code = r"""
import os, resource, signal, sys
sys.path.insert(0, %r)
from tensorflow_ocr_amd.datasets import _decode
signal.signal(signal.SIGXFSZ, signal.SIG_IGN)
resource.setrlimit(resource.RLIMIT_FSIZE, (5 << 16, 5 << 16))
slots, path = _decode._reserve_slab(16, 3, 1 << 16, "slab_c", dirs=(%r,))
print(slots, os.stat(path).st_size)
try:
    _decode._reserve_slab(16, 9, 1 << 16, "slab_d", dirs=(%r,))
except _decode.SlabUnavailable as e:
    print("unavailable", "16 slots" in str(e) and "9 slots" in str(e))
""" % (ROOT, str(tmp_path), str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.split("\n")
    assert lines[0] == "4 %d" % (4 << 16) and lines[1] == "unavailable True", r.stdout
    # end to end: a worker process writes the decoded pixels into the reserved slab
    from PIL import Image
    arr = (np.arange(40 * 50 * 3) % 251).astype(np.uint8).reshape(40, 50, 3)
    Image.fromarray(arr).save(tmp_path / "img_1.png")
    open(tmp_path / "gt_img_1.txt", "w").write("1,1,20,1,20,10,1,10,text\n")
    pool = _decode.DecodePool(1, slots=2, slot_bytes=40 * 50 * 3)
    try:
        fn, im, polys, tags, slot = pool.submit(str(tmp_path / "img_1.png"), 64).result(timeout=60)
        assert slot is not None and np.array_equal(np.asarray(im)[..., ::-1] if im.shape == arr.shape and not np.array_equal(im, arr) else im, arr)
        pool.release(slot)
    finally:
        pool.close()


_ACK_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import torch.distributed as td
from tensorflow_ocr_amd import dist, _lib as L
rank, world, _ = dist.init_process_group_from_env("gloo")
store = td.distributed_c10d._get_default_store()
if rank == 1:
    class Broken:                                   # rank 1 cannot read the id (a store.get that times out / raises)
        def __init__(self, s): self.s = s
        def get(self, k):
            if k.startswith("ocr_comm_id_") and "_ack_" not in k:
                raise RuntimeError("simulated store timeout")
            return self.s.get(k)
        def set(self, k, v): return self.s.set(k, v)
        def wait(self, *a): return self.s.wait(*a)
    store = Broken(store)
dist.AbiComm.ACK_TIMEOUT_S = 30
try:
    dist.AbiComm(rank, world, store=store)
    print("rank", rank, "entered init")             # must not happen on either rank
except (L.OcrHipError, RuntimeError) as e:
    print("rank", rank, "raised:", str(e)[:80])
td.barrier(); td.destroy_process_group()
"""


def test_rccl_rendezvous_second_gate_keeps_healthy_ranks_out_of_init(tmp_path):
    """a rank that fails to obtain the unique id AFTER the availability vote must not leave the others
    blocked inside ncclCommInitRank: every rank acknowledges the id (or its failure) through the store first, and all
    of them raise — the caller then falls back to the torch exchange collectively.  World 2, gloo, CPU: rank 1's
    store.get fails; neither rank reaches ocr_comm_init_rank and both return within the timeout."""
    script = tmp_path / "ack.py"
    script.write_text(_ACK_WORKER % ROOT)
    from tensorflow_ocr_amd import launch
    port = launch.free_port()
    ps = [subprocess.Popen([sys.executable, str(script)], env=launch.child_env(r, 2, port), stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT) for r in range(2)]
    outs = []
    for p in ps:
        try:
            outs.append(p.communicate(timeout=150)[0].decode())
        except subprocess.TimeoutExpired:
            for q in ps:
                q.kill()
            raise
    for r, (p, o) in enumerate(zip(ps, outs)):
        assert p.returncode == 0 and ("rank %d raised" % r) in o and "entered init" not in o, o[-2000:]
    assert "simulated store timeout" in outs[1] and "could not obtain the unique id" in outs[0]
