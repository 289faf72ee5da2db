"""One process per GPU, started from a plain command line.

The reference drives its towers from ONE process (`multigpu_train.py:118-133`: a `tf.device` loop
over `gpu_list`); here every tower is its own process, so `python bench.py --gpus N` /
`python multigpu_train.py --gpu_list 0,1,…` / `python train_pixellink.py --num_gpus N` must start
the other N-1 themselves when nobody (torch.distributed.run) has done it for them.

`self_launch(n)` is called FIRST THING by those scripts, before anything touches HIP:

* `WORLD_SIZE` already in the environment -> we are a rank; returns None and the script goes on.
* n <= 1 -> single tower; returns None.
* otherwise the caller is the launcher: it spawns n children running the same command line with
  RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment, relays
  rank 0's stdout (the other ranks' stdout goes to stderr, so a "one JSON line" contract holds),
  waits, and returns the exit code: 0 only if every child exited 0.  When one child fails the
  others are terminated (SIGTERM, then SIGKILL after a grace period) — they would otherwise sit in
  a collective until the RCCL watchdog fires.  The launcher never initialises the GPU and never
  exec()s: children are fresh processes.
"""
import os
import signal
import socket
import subprocess
import sys
import threading
import time


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def child_env(rank, world, port, visible=None, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // world)))
    if visible is not None:
        # the reference's os.environ['CUDA_VISIBLE_DEVICES'] = FLAGS.gpu_list (multigpu_train.py:90):
        # LOCAL_RANK then indexes into the list
        # HIP_VISIBLE_DEVICES indexes INSIDE whatever ROCR_VISIBLE_DEVICES the scheduler / container set: that
        # restriction is kept.  CUDA_VISIBLE_DEVICES is the same HIP-level filter under another name, so the
        # list replaces it.
        env["HIP_VISIBLE_DEVICES"] = visible
        env.pop("CUDA_VISIBLE_DEVICES", None)
    return env


def _reap(procs, grace=10.0):
    for p in procs:
        if p.poll() is None:
            p.send_signal(signal.SIGTERM)
    t0 = time.time()
    for p in procs:
        while p.poll() is None and time.time() - t0 < grace:
            time.sleep(0.05)
        if p.poll() is None:
            p.kill()
            p.wait()


def self_launch(nproc, argv=None, visible=None, poll=0.05):
    """See the module docstring.  `visible`: optional HIP_VISIBLE_DEVICES for the children."""
    if "WORLD_SIZE" in os.environ or nproc <= 1:
        return None
    argv = list(sys.argv if argv is None else argv)
    port = free_port()
    procs = []

    # a cancelled job (SIGTERM / SIGHUP to the launcher) must not leave N ranks waiting in a collective
    class _Cancelled(BaseException):
        pass

    def on_signal(signum, frame):
        raise _Cancelled(signum)
    old_handlers = {}
    if threading.current_thread() is threading.main_thread():
        for sg in (signal.SIGTERM, signal.SIGHUP):
            old_handlers[sg] = signal.signal(sg, on_signal)
    try:
        for r in range(nproc):
            procs.append(subprocess.Popen([sys.executable] + argv, env=child_env(r, nproc, port, visible),
                                          stdout=None if r == 0 else sys.stderr))
        rc = 0
        live = set(range(nproc))
        while live:
            for r in sorted(live):
                c = procs[r].poll()
                if c is None:
                    continue
                live.discard(r)
                if c != 0:
                    sys.stderr.write("launcher: rank %d exited with code %d; stopping the other ranks\n" % (r, c))
                    rc = c if c > 0 else 1
                    _reap(procs)
                    live.clear()
                    break
            time.sleep(poll)
        return rc
    except _Cancelled as c:
        sys.stderr.write("launcher: signal %d; stopping %d ranks\n" % (c.args[0], len(procs)))
        _reap(procs, grace=5.0)
        return 128 + int(c.args[0])
    except BaseException:
        _reap(procs, grace=2.0)
        raise
    finally:
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
