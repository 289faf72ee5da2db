"""Host side of one training sample (reference datasets/icdar.py:43-135,559-571,616-619): image decode, gt_<name>.txt
parsing, polygon validation — and the decode WORKER PROCESSES that run it (the reference's GeneratorEnqueuer with
use_multiprocessing=True, tool/data_util.py:15-128).

This module imports NumPy and PIL only (no torch, nothing of the GPU side), so a worker is a plain interpreter started as
`python -m tensorflow_ocr_amd.datasets._decode <slab> <slots> <slot_bytes>` — never a fork of the process that holds the
GPU.  Jobs and results travel as length-prefixed pickles over the worker's stdin / stdout; decoded images do not: a
worker writes the pixels into its slot of a shared slab (a file under /dev/shm that parent and workers map, unlinked as
soon as everyone has it open), and the parent copies them from there straight into the pinned upload slab.  With threads
instead (num_workers threads in the training process) the Python parts of a sample — PIL's Python-level decode loop, the
CSV parsing, the polygon checks — serialise on the GIL: 16 threads deliver ~1150 images/s on the GPU box's host where the
headline step consumes ~1650."""
import csv
import os
import pickle
import struct
import sys

import numpy as np


def load_annoataion(p):
    """icdar.py:43-66: x1,y1,...,x4,y4,label per line; label '*' or '###' = ignored text."""
    text_polys, text_tags = [], []
    if not os.path.exists(p):
        return np.array(text_polys, dtype=np.float32)
    with open(p, 'r', encoding='utf-8-sig') as f:
        for line in csv.reader(f):
            if not line:
                continue
            label = line[-1]
            line = [i.strip('\ufeff').strip('\xef\xbb\xbf') for i in line]
            x1, y1, x2, y2, x3, y3, x4, y4 = list(map(float, line[:8]))
            text_polys.append([[x1, y1], [x2, y2], [x3, y3], [x4, y4]])
            text_tags.append(label == '*' or label == '###')
    return np.array(text_polys, dtype=np.float32), np.array(text_tags, dtype=bool)


def polygon_area(poly):
    """icdar.py:69-81 (shoelace, sign = orientation)."""
    edge = [(poly[1][0] - poly[0][0]) * (poly[1][1] + poly[0][1]),
            (poly[2][0] - poly[1][0]) * (poly[2][1] + poly[1][1]),
            (poly[3][0] - poly[2][0]) * (poly[3][1] + poly[2][1]),
            (poly[0][0] - poly[3][0]) * (poly[0][1] + poly[3][1])]
    return np.sum(edge) / 2.


def check_and_validate_polys(polys, tags, size):
    """icdar.py:108-135: clip to the image, drop |area| < 1, flip clockwise-wrong polygons."""
    (h, w) = size
    if polys.shape[0] == 0:
        return polys
    polys[:, :, 0] = np.clip(polys[:, :, 0], 0, w - 1)
    polys[:, :, 1] = np.clip(polys[:, :, 1], 0, h - 1)
    validated_polys, validated_tags = [], []
    for poly, tag in zip(polys, tags):
        p_area = polygon_area(poly)
        if abs(p_area) < 1:
            continue
        if p_area > 0:
            poly = poly[(0, 3, 2, 1), :]
        validated_polys.append(poly)
        validated_tags.append(tag)
    return np.array(validated_polys), np.array(validated_tags)


def read_image_rgb(path):
    """cv2.imread(...)[:, :, ::-1]: uint8 [H,W,3] RGB.  .npy arrays are taken as RGB already."""
    if path.endswith('.npy'):
        return np.ascontiguousarray(np.load(path), dtype=np.uint8)
    from PIL import Image
    with Image.open(path) as im:
        if im.mode != "RGB":
            im = im.convert("RGB")
        return np.asarray(im, dtype=np.uint8)


def txt_name(im_fn):
    """icdar.py:564: <dir>/gt_<stem>.txt"""
    return im_fn[:im_fn.rfind('/') + 1] + 'gt_' + im_fn[im_fn.rfind('/') + 1:im_fn.rfind('.')] + '.txt'


def load_sample(args):
    """One sample (icdar.py:559-571,616-619): decode, parse, validate, scale the polygons to the training size.
    Returns None for a sample the reference skips, else (im_fn, image uint8 [H,W,3], polys float32 [k,4,2], tags bool [k])."""
    im_fn, input_size = args
    tf = txt_name(im_fn)
    if not os.path.exists(tf):
        return None
    try:
        im = read_image_rgb(im_fn)
        h, w, _ = im.shape
        text_polys, text_tags = load_annoataion(tf)
        text_polys, text_tags = check_and_validate_polys(text_polys, text_tags, (h, w))
        if text_polys.shape[0] == 0:
            return None
        text_polys[:, :, 0] *= input_size / float(w)
        text_polys[:, :, 1] *= input_size / float(h)
    except Exception:                       # the reference prints the traceback and moves on (:646-649)
        import traceback
        traceback.print_exc()
        return None
    return im_fn, im, text_polys, text_tags


# ----------------------------------------------------------------------------------- worker processes
def _send(f, obj):
    b = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    f.write(struct.pack("<Q", len(b)))
    f.write(b)
    f.flush()


def _recv(f):
    hdr = f.read(8)
    if len(hdr) < 8:
        return None
    (n,) = struct.unpack("<Q", hdr)
    b = f.read(n)
    if len(b) < n:
        return None
    return pickle.loads(b)


def _worker_main(path, slots, slot_bytes):
    fin, fout = sys.stdin.buffer, sys.stdout.buffer
    sys.stdout = sys.stderr                  # anything printed by the sample code goes to stderr, not into the protocol
    slab = np.memmap(path, dtype=np.uint8, mode="r+", shape=(slots, slot_bytes))
    _send(fout, "ready")
    while True:
        job = _recv(fin)
        if job is None:
            return
        im_fn, input_size, slot = job
        smp = load_sample((im_fn, input_size))
        if smp is None:
            _send(fout, None)
            continue
        fn, im, polys, tags = smp
        if im.nbytes <= slot_bytes:
            slab[slot, :im.nbytes] = im.reshape(-1)
            _send(fout, (fn, tuple(im.shape), None, polys, tags))
        else:                                # larger than a slot: the pixels travel through the pipe
            _send(fout, (fn, tuple(im.shape), im, polys, tags))


class SlabUnavailable(RuntimeError):
    """No directory could hold the decode slab (datasets.icdar.generator then decodes on threads)."""


def _reserve_slab(slots, min_slots, slot_bytes, name, dirs=None):
    """Create the slab file with its pages RESERVED (posix_fallocate), so that a full /dev/shm is an error here and not
    a SIGBUS at the first touch of a page past the limit (np.memmap(mode="w+") succeeds on a sparse file whatever the
    file system holds: Docker's default /dev/shm is 64 MB, a batch-32 slab 189 MB).  Tries /dev/shm with `slots`, then
    with fewer (down to `min_slots`: a batch's images are held at once), then the temp directory (page cache instead
    of shared memory: same semantics).  Returns (slots, path); raises SlabUnavailable with what was tried."""
    import errno
    import tempfile
    tried = []
    for d in (dirs if dirs is not None else ("/dev/shm", tempfile.gettempdir())):
        if not os.path.isdir(d) or not os.access(d, os.W_OK | os.X_OK):
            tried.append("%s: missing or not writable" % d)
            continue
        n = slots
        while n >= min_slots:
            path = os.path.join(d, name)
            try:
                fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
            except OSError as e:
                tried.append("%s: %s" % (path, e.strerror))
                break
            try:
                os.posix_fallocate(fd, 0, n * slot_bytes)
                return n, path
            except OSError as e:
                os.unlink(path)
                tried.append("%s: %d slots (%d MB): %s" % (d, n, n * slot_bytes >> 20, e.strerror))
                if e.errno not in (errno.ENOSPC, errno.EFBIG, errno.EDQUOT) or n == min_slots:
                    break
                n = max(min_slots, n // 2)
            finally:
                os.close(fd)
    raise SlabUnavailable("no room for the decode slab: " + "; ".join(tried))


class DecodePool:
    """`workers` decode processes + a slab of `slots` image slots shared with them.  submit() returns a Future whose
    result is None (sample skipped) or (im_fn, image, polys, tags) with `image` a uint8 [H,W,3] VIEW of the sample's slot:
    copy it out (datasets.icdar.resize_images does, into the pinned slab) and then release(slot)."""

    def __init__(self, workers, slots=None, slot_bytes=3 * 1280 * 768, min_slots=None):
        import queue
        import subprocess
        import threading
        self.workers = workers
        self.slot_bytes = int(slot_bytes)
        self.slots, self.path = _reserve_slab(slots or 4 * workers, min_slots or min(slots or 4 * workers, 2 * workers),
                                              self.slot_bytes, "ocr_decode_%d_%x" % (os.getpid(), id(self)))
        self.slab = np.memmap(self.path, dtype=np.uint8, mode="r+", shape=(self.slots, self.slot_bytes))
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        env = dict(os.environ)
        env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
        for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
            env[k] = "1"
        self.procs = []
        try:
            for _ in range(workers):
                self.procs.append(subprocess.Popen(
                    [sys.executable, "-m", "tensorflow_ocr_amd.datasets._decode", self.path, str(self.slots), str(self.slot_bytes)],
                    stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env, cwd=root))
            for p in self.procs:
                if _recv(p.stdout) != "ready":
                    raise RuntimeError("decode worker failed to start")
        finally:
            try:
                os.unlink(self.path)         # everyone has it mapped: nothing is left behind whatever happens next
            except OSError:
                pass
        self.free = queue.Queue()
        for s in range(self.slots):
            self.free.put(s)
        self.jobs = queue.Queue()
        self.closed = False
        self.threads = [threading.Thread(target=self._serve, args=(p,), daemon=True) for p in self.procs]
        for t in self.threads:
            t.start()

    def _serve(self, proc):
        while True:
            item = self.jobs.get()
            if item is None:
                return
            fut, im_fn, input_size, slot = item
            try:
                _send(proc.stdin, (im_fn, input_size, slot))
                res = _recv(proc.stdout)
                if res is None:
                    self.free.put(slot)
                    fut.set_result(None)
                    continue
                fn, shape, im, polys, tags = res
                if im is None:
                    nb = int(np.prod(shape))
                    im = self.slab[slot, :nb].reshape(shape)
                    fut.set_result((fn, im, polys, tags, slot))
                else:
                    self.free.put(slot)
                    fut.set_result((fn, im, polys, tags, None))
            except Exception as e:            # broken pipe, worker died
                fut.set_exception(e)
                return

    def submit(self, im_fn, input_size):
        """Blocks while every slot is in use (back-pressure: at most `slots` decoded images wait for the consumer)."""
        from concurrent.futures import Future
        slot = self.free.get()
        fut = Future()
        self.jobs.put((fut, im_fn, input_size, slot))
        return fut

    def release(self, slot):
        if slot is not None:
            self.free.put(slot)

    def close(self):
        if self.closed:
            return
        self.closed = True
        for _ in self.threads:
            self.jobs.put(None)
        for p in self.procs:
            try:
                p.stdin.close()
            except Exception:
                pass
        for p in self.procs:
            try:
                p.wait(timeout=5)
            except Exception:
                p.kill()
        self.slab = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


if __name__ == "__main__":
    _worker_main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
