"""Mirror of the reference's datasets/icdar.py on the training path: ICDAR
annotation parsing and validation on the host (tiny), label maps and image preparation on the GPU.

Same function names and argument order as the reference where it has them:
  load_annoataion           icdar.py:43-66   gt_<name>.txt CSV -> (polys float32 [k,4,2], tags bool [k])
  polygon_area              icdar.py:69-81
  check_and_validate_polys  icdar.py:108-135
  generate_rbox             icdar.py:486-539 (one image, full resolution, NumPy results)
  generator / get_batch     icdar.py:542-668 (resize to input_size, BGR->RGB float, labels at 1/4)
plus the batched device entry point the feeder uses:
  generate_rbox_batch       all images of a batch in two launches; device tensors at 1/4 resolution

The reference rasterises with cv2.fillPoly and resizes with cv2.resize; here both run as HIP kernels
(ocr_poly_cover / ocr_icdar_labels / ocr_resize_linear_u8) that reproduce OpenCV's rasters bit for bit
(tests/test_gpu_labels.py).  There is no CPU fallback.
"""
import csv
import glob
import os

import numpy as np
import torch

from .. import ops
from ..graph import F32, get_default_graph

MIN_TEXT_SIZE = 10          # FLAGS.min_text_size (icdar.py:24)
MAX_POLYS = 254             # poly_mask is a uint8 image in the reference


def get_images(training_data_path):
    """icdar.py:36-41 (plus .npy arrays for containers without an image decoder)."""
    files = []
    for ext in ['jpg', 'png', 'jpeg', 'JPG', 'npy']:
        files.extend(glob.glob(os.path.join(training_data_path, '*.{}'.format(ext))))
    return files


# host side of a sample: datasets/_decode.py (NumPy + PIL only, so that decode WORKER PROCESSES can import it)
from ._decode import (check_and_validate_polys, load_annoataion, polygon_area, read_image_rgb, txt_name,  # noqa: E402,F401
                      load_sample as _load_sample)


def _ignore_flags(polys, tags, min_text_size):
    """icdar.py:509-515: the polygon is zeroed in the training mask when it is small or tagged."""
    out = np.zeros(len(polys), np.uint8)
    for i, (poly, tag) in enumerate(zip(polys, tags)):
        poly_h = min(np.linalg.norm(poly[0] - poly[3]), np.linalg.norm(poly[1] - poly[2]))
        poly_w = min(np.linalg.norm(poly[0] - poly[1]), np.linalg.norm(poly[2] - poly[3]))
        out[i] = 1 if (min(poly_h, poly_w) < min_text_size or tag) else 0
    return out


def pack_polys(polys_list, tags_list, min_text_size=MIN_TEXT_SIZE):
    """Host packing for the kernels: int32 vertices (`poly.astype(np.int32)`, icdar.py:505),
    per-image counts and ignore flags.  Returns numpy (polys [n,P,4,2], counts [n], ignore [n,P])."""
    n = len(polys_list)
    P = max([len(p) for p in polys_list] + [1])
    if P > MAX_POLYS:
        raise ValueError("at most %d polygons per image (poly_mask is uint8 in the reference)" % MAX_POLYS)
    polys = np.zeros((n, P, 4, 2), np.int32)
    counts = np.zeros(n, np.int32)
    ignore = np.zeros((n, P), np.uint8)
    for b, (pl, tg) in enumerate(zip(polys_list, tags_list)):
        pl = np.asarray(pl, np.float32).reshape(-1, 4, 2)
        counts[b] = len(pl)
        if len(pl):
            polys[b, :len(pl)] = pl.astype(np.int32)
            ignore[b, :len(pl)] = _ignore_flags(pl, tg, min_text_size)
    return polys, counts, ignore


def generate_rbox_batch(im_size, polys_list, tags_list, step=4, min_text_size=MIN_TEXT_SIZE, graph=None):
    """generate_rbox for every image of a batch + the generator's `[::step, ::step]` subsample and
    float casts (icdar.py:632-634).  Returns device tensors score [n,h/s,w/s,1], geo [n,h/s,w/s,8],
    training mask [n,h/s,w/s,1] (float32) — the feeds `input_score_maps`, `input_geo_maps`,
    `input_training_masks` of multigpu_train.py:98-101."""
    g = graph or get_default_graph()
    h, w = im_size
    if h != w:
        raise ValueError("generate_rbox needs a square image: the reference's valid_link border rule "
                         "(icdar.py:84) indexes out of range otherwise")
    polys, counts, ignore = pack_polys(polys_list, tags_list, min_text_size)
    n = len(counts)
    dev = g.device
    d_polys = torch.from_numpy(polys).to(dev)
    d_counts = torch.from_numpy(counts).to(dev)
    d_ignore = torch.from_numpy(ignore).to(dev)
    cover = torch.empty((n, h, w), dtype=torch.int32, device=dev)
    ops.poly_cover(d_polys, d_counts, d_ignore, h, w, cover)
    oh, ow = -(-h // step), -(-w // step)
    score = torch.empty((n, oh, ow, 1), dtype=F32, device=dev)
    geo = torch.empty((n, oh, ow, 8), dtype=F32, device=dev)
    mask = torch.empty((n, oh, ow, 1), dtype=F32, device=dev)
    ops.icdar_labels(cover, step, score, geo, mask)
    return score, geo, mask


def generate_rbox(im_size, polys, tags, min_text_size=MIN_TEXT_SIZE, graph=None):
    """icdar.py:486-539 for one image: (score_map uint8 [h,w], geo_map float32 [h,w,8], training_mask
    uint8 [h,w]) as NumPy arrays at full resolution."""
    s, gmap, m = generate_rbox_batch(im_size, [polys], [tags], step=1, min_text_size=min_text_size, graph=graph)
    return (s[0, :, :, 0].cpu().numpy().astype(np.uint8), gmap[0].cpu().numpy(),
            m[0, :, :, 0].cpu().numpy().astype(np.uint8))


def resize_images(images_u8, input_size, graph=None):
    """`cv2.resize(im, dsize=(input_size, input_size))` + `.astype(np.float32)` (icdar.py:615,630) for a
    list of uint8 [H,W,3] images of any size -> device float32 [n,S,S,3].  The batch crosses PCIe
    once: the images are packed into ONE pinned slab, copied asynchronously on the current stream
    (torch's pinned-memory cache holds the slab until that copy has completed) and resized from
    views into the device copy."""
    g = graph or get_default_graph()
    out = torch.empty((len(images_u8), input_size, input_size, 3), dtype=F32, device=g.device)
    sizes = [int(np.prod(im.shape)) for im in images_u8]
    if not sizes:
        return out
    slab = torch.empty(sum(sizes), dtype=torch.uint8, pin_memory=(g.device.type == "cuda"))
    host = slab.numpy()
    off = 0
    for im, sz in zip(images_u8, sizes):
        host[off:off + sz] = np.asarray(im, dtype=np.uint8).reshape(-1)
        off += sz
    dev = slab.to(g.device, non_blocking=True)
    off = 0
    for b, (im, sz) in enumerate(zip(images_u8, sizes)):
        ops.resize_linear_u8(dev[off:off + sz].view(im.shape), out[b])
        off += sz
    return out


def generator(training_data_path, input_size=512, batch_size=32, graph=None, shuffle=True, seed=None,
              num_workers=0, worker_kind=None):
    """icdar.py:542-649 with the branches the reference has live (no random scale / crop: `if (0)`):
    read image + gt, validate polygons, resize to input_size x input_size, scale the polygons,
    labels at 1/4 resolution.  Yields (images [B,S,S,3] float32 RGB, image_fns, score_maps, geo_maps,
    training_masks) as DEVICE tensors (the reference yields lists of NumPy arrays).  num_workers > 0:
    that many decode workers run ahead (the reference's GeneratorEnqueuer workers) — worker_kind "process" (default;
    OCR_DECODE_WORKERS): plain interpreters that import NumPy + PIL only and hand the pixels over through a shared slab
    (datasets/_decode.py: DecodePool); "thread": threads of this process (their Python parts serialise on the GIL)."""
    image_list = np.array(sorted(get_images(training_data_path)))
    print('{} training images in {}'.format(image_list.shape[0], training_data_path))
    if len(image_list) == 0:
        return
    index = np.arange(0, image_list.shape[0])
    rng = np.random.RandomState(seed)
    kind = worker_kind or os.environ.get("OCR_DECODE_WORKERS", "process")
    if kind not in ("process", "thread"):
        raise ValueError("worker_kind must be 'process' or 'thread'")
    pool = dpool = None
    if num_workers > 0 and kind == "thread":
        from multiprocessing.pool import ThreadPool
        pool = ThreadPool(num_workers)
    elif num_workers > 0:
        from ._decode import DecodePool, SlabUnavailable
        try:
            # a batch's images sit in their slots at once: at least batch_size + 1 slots (one ahead), ideally two batches
            dpool = DecodePool(num_workers, slots=max(2 * batch_size, 2 * num_workers),
                               min_slots=max(batch_size + 1, num_workers + 1))
        except SlabUnavailable as e:
            import warnings
            warnings.warn("decode worker processes need a shared slab and found no room for it (%s): decoding on %d "
                          "threads instead (slower: their Python parts share the GIL)" % (e, num_workers))
            from multiprocessing.pool import ThreadPool
            pool = ThreadPool(num_workers)

    def samples_of(jobs):
        """The epoch's samples in job order: (im_fn, image, polys, tags, slot | None), None for skipped ones."""
        if dpool is None:
            for smp in (pool.imap(_load_sample, jobs, chunksize=4) if pool else map(_load_sample, jobs)):
                yield None if smp is None else smp + (None,)
            return
        from collections import deque
        pending, it = deque(), iter(jobs)
        ahead = max(1, dpool.slots - batch_size)           # leave a batch worth of slots to the images being consumed
        while True:
            while len(pending) < ahead:
                j = next(it, None)
                if j is None:
                    break
                pending.append(dpool.submit(*j))
            if not pending:
                return
            yield pending.popleft().result()
    try:
        while True:
            if shuffle:
                rng.shuffle(index)
            jobs = [(str(image_list[i]), input_size) for i in index]
            ims, fns, polys_l, tags_l, slots = [], [], [], [], []
            produced = False
            for smp in samples_of(jobs):
                if smp is None:
                    continue
                fns.append(smp[0])
                ims.append(smp[1])
                polys_l.append(smp[2])
                tags_l.append(smp[3])
                slots.append(smp[4])
                if len(ims) == batch_size:
                    images = resize_images(ims, input_size, graph=graph)      # copies the pixels out of their slots
                    for sl in slots:
                        if dpool is not None:
                            dpool.release(sl)
                    score, geo, mask = generate_rbox_batch((input_size, input_size), polys_l, tags_l, graph=graph)
                    produced = True
                    yield images, fns, score, geo, mask
                    ims, fns, polys_l, tags_l, slots = [], [], [], [], []
            for sl in slots:                                                  # the epoch's incomplete last batch
                if dpool is not None:
                    dpool.release(sl)
            if not produced:
                return                       # fewer usable samples than one batch: do not spin
    finally:
        if pool is not None:
            pool.terminate()
            pool.join()
        if dpool is not None:
            dpool.close()


def get_batch(num_workers=0, device_prefetch=2, **kwargs):
    """icdar.py:652-668: background producer + queue.  Host decode/parsing runs in `num_workers`
    threads; upload, resize and label kernels run on the feeder's own HIP stream
    (feeder.DeviceFeeder), `device_prefetch` batches ahead of the training step."""
    from ..feeder import DeviceFeeder
    graph = kwargs.get("graph") or get_default_graph()
    return DeviceFeeder(lambda: generator(num_workers=num_workers, **kwargs), graph.device, depth=device_prefetch)
