#!/usr/bin/env python3
"""Measurement of the BASELINE.json configs other than the headline one (which is bench.py's):

  pixellink  configs[2]: PixelLink VGG-16 512x512 batch 32 train step (softmax/OHNM + focal-style link
             loss, Momentum) followed by the link-CC decode of the same batch
  resnet     configs[3] per-GPU share: EAST ResNet-v1-50 640x640 batch 64 train step (dice), 1 GPU
  decode     configs[4]: PixelLink inference 1024x1024 batch 16, then pixel/link softmax + link-CC
             decode, and locality-aware NMS on synthetic quadrangle lists

  east_fwd   configs[0]: the test.py path on ONE 512x512 image — model.model (ResNet-v1-50 + PixelLink heads,
             is_training=False), softmaxes, pixel_detect twin, contour boxes — with the CPU restatement's
             forward timed beside it (the reference runs this config on TF-CPU)
  pipeline   SURVEY 8f-1/2/4: label maps for 32 x 512^2 (16 quads each), cv2.resize of 32 720x1280
             images to 512^2, oriented boxes of the decode's components; with the CPU restatement timed
             beside them (baseline only)

Prints one JSON line per config.  Same timing discipline as bench.py (resident inputs, warm-up,
synchronize on both sides).  Dev/measurement tool: not part of the driver contract."""
import argparse
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))


def timed(fn, warmup, steps):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


MFMA_PEAK_TFLOPS = 2500.0          # dense f16 / bf16 MFMA, MI355X_MICROARCH.md (same constant as bench.py)


def dominant_kernel(timing):
    """(name, TFLOP/s, launches) of the conv instantiation with the most accumulated HIP-event time."""
    per = {}
    for variant, flops, phase, e0, e1 in timing:
        a = per.setdefault(variant, [0.0, 0.0, 0])
        a[0] += flops
        a[1] += e0.elapsed_time(e1) * 1e-3
        a[2] += 1
    if not per:
        return None
    k = max(per, key=lambda n: per[n][1])
    fl, sec, cnt = per[k]
    return {"kernel": k, "achieved_tflops": round(fl / sec / 1e12, 1), "frac": round(fl / sec / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "launches": cnt, "avg_launch_ms": round(sec / cnt * 1e3, 4)}


def train_config(name, forward_loss, opt_factory, batch, size, steps, warmup, extra=None, prep=None, gflop_per_img=None):
    from tensorflow_ocr_amd import _lib, ops, synthetic
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.train import TrainStep
    dev = torch.device("cuda", 0)
    g = Graph(dev, loss_scale=1024.0, seed=1)
    rng = np.random.default_rng(100)
    data = [torch.from_numpy(a).to(dev) for a in synthetic.make_batch(rng, batch, size)]
    if prep is not None:
        data = prep(data)         # input-pipeline work (normalisation): before the recorded step, never inside it
    step = TrainStep(g, forward_loss, opt_factory)
    loss = None
    for _ in range(3):
        loss = step(*data)
    for _ in range(warmup):
        step(*data)
    torch.cuda.synchronize()
    ops.KERNEL_TIMING = []
    t0 = time.perf_counter()
    for _ in range(steps):
        step(*data)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    timing, ops.KERNEL_TIMING = ops.KERNEL_TIMING, None
    out = {"config": name, "batch": batch, "size": size, "steps": steps, "dtype": _lib.STORAGE,
           "ms_per_step": round(dt * 1e3, 3),
           "images_per_sec": round(batch / dt, 1), "loss": round(float(step(*data).item()), 5),
           "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}
    if gflop_per_img:
        tf = batch * gflop_per_img / 1e3 / dt
        out["tflops"] = round(tf, 1)
        out["frac_of_peak"] = round(tf / MFMA_PEAK_TFLOPS, 4)
    out["dominant_kernel"] = dominant_kernel(timing)
    if extra:
        out.update(extra(g, data))
    print(json.dumps(out), flush=True)
    return g


def make_icdar_dir(root, count=128, h=720, w=1280, quads=16, seed=0):
    """A synthetic ICDAR-2015-shaped training directory, generated once: `count` 720 x 1280 JPEGs (smooth background +
    text-like boxes: ~100-200 KB each, the size class of the real set) and their gt_<name>.txt (x1,y1,..,x4,y4,label)."""
    import os
    from PIL import Image
    os.makedirs(root, exist_ok=True)
    if len([f for f in os.listdir(root) if f.endswith(".jpg")]) >= count:
        return root
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    bases = []
    for j in range(4):                        # four smooth backgrounds with sensor-like noise, re-used shifted
        a, b, c = rng.uniform(0.5, 2.0, 3)
        bg = np.stack([127 + 100 * np.sin(xx / w * 6.28 * a + k) * np.cos(yy / h * 6.28 * b * (k + 1) / 2 + c) for k in range(3)], -1)
        bases.append(np.clip(bg + rng.normal(0, 6, bg.shape), 0, 255).astype(np.uint8))
    for i in range(count):
        base = np.roll(bases[i % 4], (int(rng.integers(0, h)), int(rng.integers(0, w))), axis=(0, 1)).copy()
        lines = []
        for q in range(quads):
            bw, bh = int(rng.integers(60, 260)), int(rng.integers(18, 60))
            x0, y0 = int(rng.integers(4, w - bw - 4)), int(rng.integers(4, h - bh - 4))
            base[y0:y0 + bh, x0:x0 + bw] = rng.integers(0, 255, 3)
            base[y0 + bh // 4:y0 + 3 * bh // 4:2, x0 + 4:x0 + bw - 4:3] = rng.integers(0, 255, 3)
            lab = "###" if q % 8 == 7 else "text%d" % q
            lines.append("%d,%d,%d,%d,%d,%d,%d,%d,%s" % (x0, y0, x0 + bw, y0, x0 + bw, y0 + bh, x0, y0 + bh, lab))
        Image.fromarray(base).save(os.path.join(root, "img_%d.jpg" % (i + 1)), quality=90)
        with open(os.path.join(root, "gt_img_%d.txt" % (i + 1)), "w") as f:
            f.write("\r\n".join(lines) + "\r\n")
    return root


def pipeline_train(steps, warmup, batch=32, size=512, workers=16, root="/tmp/ocr_icdar_synth"):
    """multigpu_train.py's hot loop (:164-174) WITH its input path — icdar.get_batch (:652-668: files
    on disk -> decode + parse in `workers` host threads -> pinned slab -> cv2.resize / label-map kernels on the feeder's
    stream) feeding the recorded model_vgg + dice step — against the same step on a resident batch."""
    import os
    from tensorflow_ocr_amd import _lib
    from tensorflow_ocr_amd.datasets import icdar
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import model_vgg_16 as M
    from tensorflow_ocr_amd.train import AdamOptimizer, TrainStep
    t0 = time.perf_counter()
    root = make_icdar_dir(root)
    gen_s = time.perf_counter() - t0
    files = icdar.get_images(root)
    jpg_kb = float(np.mean([os.path.getsize(f) for f in files[:64]])) / 1024.0
    # host stage alone: decode + annotation parsing + polygon validation in the workers, nothing on the GPU — with the
    # worker kind the feeder will use (OCR_DECODE_WORKERS: processes by default, threads on request)
    kind = os.environ.get("OCR_DECODE_WORKERS", "process")
    jobs = [(f, size) for f in files] * 2
    if kind == "thread":
        from multiprocessing.pool import ThreadPool
        pool = ThreadPool(workers)
        list(pool.imap(icdar._load_sample, jobs[:2 * workers], chunksize=4))
        t0 = time.perf_counter()
        n_ok = sum(1 for smp in pool.imap(icdar._load_sample, jobs, chunksize=4) if smp is not None)
        host_rate = n_ok / (time.perf_counter() - t0)
        pool.terminate()
        pool.join()
    else:
        from collections import deque
        from tensorflow_ocr_amd.datasets._decode import DecodePool
        dp = DecodePool(workers, slots=4 * workers)
        for rnd in range(2):                  # round 0 warms the page cache and the workers
            pend, it, n_ok = deque(), iter(jobs), 0
            t0 = time.perf_counter()
            while True:
                while len(pend) < 3 * workers:
                    j = next(it, None)
                    if j is None:
                        break
                    pend.append(dp.submit(*j))
                if not pend:
                    break
                r = pend.popleft().result()
                if r is not None:
                    n_ok += 1
                    dp.release(r[4])
            host_rate = n_ok / (time.perf_counter() - t0)
        dp.close()
    dev = torch.device("cuda", 0)
    g = Graph(dev, loss_scale=1024.0, seed=1)

    def fl(gr, im, sm, gm, tm):
        a, b = M.model_vgg(im, is_training=True, graph=gr)
        return M.loss(sm, a, gm, b, tm, graph=gr)
    step = TrainStep(g, fl, lambda gr: AdamOptimizer(gr, learning_rate=1e-4))
    feeder = icdar.get_batch(num_workers=workers, training_data_path=root, input_size=size, batch_size=batch, graph=g, seed=1)
    try:
        def nxt():
            images, _, score, geo, mask = next(feeder)
            return [images, score, geo, mask]
        first = nxt()
        for _ in range(3):
            step(*first)
        for _ in range(warmup):
            step(*first)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(*first)
        torch.cuda.synchronize()
        resident = (time.perf_counter() - t0) / steps
        for _ in range(warmup):
            step(*nxt())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        waits = 0.0
        for _ in range(steps):
            tw = time.perf_counter()
            b = nxt()
            waits += time.perf_counter() - tw
            loss = step(*b)
        torch.cuda.synchronize()
        fed = (time.perf_counter() - t0) / steps
    finally:
        feeder.close()
    need = batch / resident
    out = {"config": "multigpu_train hot loop with its input path: %d x 720x1280 synthetic JPEG (%.0f KB) + gt_*.txt, files in "
                     "the PAGE CACHE (the set is cycled; storage is not measured) -> "
                     "icdar.get_batch (%d decode %s workers, DeviceFeeder) -> model_vgg + dice + Adam/EMA, batch %d at %d^2"
                     % (len(files), jpg_kb, workers, kind, batch, size),
           "batch": batch, "size": size, "steps": steps, "dtype": _lib.STORAGE,
           "ms_per_step": round(fed * 1e3, 3), "images_per_sec": round(batch / fed, 1),
           "step_ms_fed": round(fed * 1e3, 3), "step_ms_resident": round(resident * 1e3, 3),
           "fed_over_resident": round(fed / resident, 4),
           "host_decode_img_s": round(host_rate, 1), "host_workers": workers, "host_worker_kind": kind, "host_cpus": os.cpu_count(),
           "host_wait_ms_per_step": round(waits / steps * 1e3, 3),
           "limiting_stage": ("none: the fed step is within 3 % of the resident step" if fed <= 1.03 * resident else
                              ("host decode + parse: %.0f images/s on %d %s workers against the %.0f images/s the resident step takes"
                               % (host_rate, workers, kind, need) if host_rate < 1.1 * need else
                               "feeder hand-over (device upload / resize / label kernels sharing the chip with the step)")),
           "files_in_page_cache": True,
           "caveat": "128 synthetic images (4 backgrounds shifted + boxes) read from the page cache: decode cost of real ICDAR "
                     "photos and storage bandwidth are not represented",
           "loss": round(float(loss.item()), 5), "dataset_generation_s": round(gen_s, 1)}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="pixellink,resnet,decode")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pixellink-loss", default="focal", choices=("focal", "plain"))
    ap.add_argument("--resnet-batch", type=int, default=64)
    ap.add_argument("--resnet-size", type=int, default=640)
    args = ap.parse_args()
    which = args.which.split(",")
    from tensorflow_ocr_amd.train import AdamOptimizer, MomentumOptimizer

    if "pipeline_train" in which:
        pipeline_train(max(args.steps, 30), max(args.warmup, 3))

    if "pixellink" in which:
        from tensorflow_ocr_amd.nets import pixellink
        from tensorflow_ocr_amd.tool import pixellink_fn

        def fl(g, im, sm, gm, tm):
            net = pixellink.PixelLinkNet(im, graph=g)          # preprocessed input (prep below)
            fl.net = net
            # BASELINE.json configs[2] words the loss "softmax + focal link loss": focal=(alpha, gamma) swaps the link CE
            # for the focal form (absent in the reference, build-defined); --pixellink-loss plain = the
            # reference's own build_loss (nets/pixellink.py:88-263)
            return net.build_loss(sm[..., 0], gm, focal=(0.25, 2.0) if args.pixellink_loss == "focal" else None)

        def decode(g, data):
            from tensorflow_ocr_amd.graph import Graph
            g2 = Graph(torch.device("cuda", 0))
            n = 32
            rng = np.random.default_rng(7)
            pix = torch.from_numpy(rng.normal(0, 2, (n, 128, 128, 2)).astype(np.float32)).cuda()
            lnk = torch.from_numpy(rng.normal(0, 2, (n, 128, 128, 16)).astype(np.float32)).cuda()

            def run():
                ps = pixellink_fn.pixel_scores(pix, graph=g2)
                ls = pixellink_fn.link_scores(lnk, graph=g2)
                pixellink_fn.link_cc_decode(ps[..., 1].contiguous(), ls, 0.6, 0.6, min_size=10, graph=g2)
            dt = timed(run, 2, 10)
            return {"decode_ms_per_batch": round(dt * 1e3, 3), "decode_images_per_sec": round(n / dt, 1)}
        train_config("PixelLink VGG-16 512x512 b32: softmax pixel CE + %s link loss, Momentum" % (
            "focal (alpha 0.25, gamma 2)" if args.pixellink_loss == "focal" else "plain CE"), fl,
                     lambda gr: MomentumOptimizer(gr), 32, 512, args.steps, args.warmup, extra=decode,
                     prep=lambda d: [(d[0] - 120.0) / 60.0] + d[1:], gflop_per_img=516.5)

    if "resnet" in which:
        from tensorflow_ocr_amd.nets import model_vgg_16 as MV

        def fr(g, im, sm, gm, tm):
            a, b = MV.model(im, is_training=True, graph=g)
            return MV.loss(sm, a, gm, b, tm, graph=g)
        train_config("EAST ResNet-v1-50 %d^2 b%d: dice, Adam+EMA (per-GPU share of configs[3])" % (
            args.resnet_size, args.resnet_batch), fr, lambda gr: AdamOptimizer(gr, learning_rate=1e-4),
            args.resnet_batch, args.resnet_size, args.steps, args.warmup,
            gflop_per_img=180.9 * (args.resnet_size / 640.0) ** 2)      # 60.3 GFLOP/img forward x 3

    if "decode" in which:
        from tensorflow_ocr_amd.graph import Graph
        from tensorflow_ocr_amd.nets import pixellink
        from tensorflow_ocr_amd.tool import lanms, pixellink_fn
        dev = torch.device("cuda", 0)
        g = Graph(dev)
        rng = np.random.default_rng(3)
        n, S = 16, 1024
        x = torch.from_numpy(rng.uniform(0, 255, (n, S, S, 3)).astype(np.float32)).to(dev)
        xin = (x - 120.0) / 60.0
        holder = {}

        def fwd():
            net = pixellink.PixelLinkNet(xin, graph=g)
            g.reset_tape()
            holder["net"] = net
        fwd()
        from tensorflow_ocr_amd import ops as _ops
        fwd()
        torch.cuda.synchronize()
        _ops.KERNEL_TIMING = []
        dt_net = timed(fwd, 0, 3)
        timing, _ops.KERNEL_TIMING = _ops.KERNEL_TIMING, None
        net = holder["net"]
        rng2 = np.random.default_rng(4)
        pix = torch.from_numpy(rng2.normal(0, 2, (n, S // 4, S // 4, 2)).astype(np.float32)).to(dev)
        lnk = torch.from_numpy(rng2.normal(0, 2, (n, S // 4, S // 4, 16)).astype(np.float32)).to(dev)
        res = {}

        def dec():
            ps = pixellink_fn.pixel_scores(pix, graph=g)
            ls = pixellink_fn.link_scores(lnk, graph=g)
            res["out"] = pixellink_fn.link_cc_decode(ps[..., 1].contiguous(), ls, 0.6, 0.6, min_size=10, graph=g)
        dt_dec = timed(dec, 2, 10)
        ncomp = int(res["out"][1].sum().item())
        # LANMS on synthetic row-major quad lists (1024 candidates per image)
        K = 1024
        boxes = np.zeros((n, K, 9), np.float32)
        for i in range(n):
            cx = np.sort(rng2.uniform(20, S - 20, K)); cy = rng2.uniform(20, S - 20, K)
            w = rng2.uniform(20, 60, K); h = rng2.uniform(10, 30, K)
            boxes[i, :, 0] = cx - w; boxes[i, :, 1] = cy - h; boxes[i, :, 2] = cx + w; boxes[i, :, 3] = cy - h
            boxes[i, :, 4] = cx + w; boxes[i, :, 5] = cy + h; boxes[i, :, 6] = cx - w; boxes[i, :, 7] = cy + h
            boxes[i, :, 8] = rng2.uniform(0.5, 1.0, K)
        bt = torch.from_numpy(boxes).to(dev)
        ct = torch.full((n,), K, dtype=torch.int32, device=dev)
        dt_nms = timed(lambda: lanms.lanms_batch(bt, ct, 0.2, graph=g), 2, 10)
        px = n * (S // 4) ** 2
        fwd_tf = n * 2 * 344.9 / 1e3 / dt_net              # 344.9 GMAC per 1024^2 image
        print(json.dumps({"config": "PixelLink inference 1024^2 b16 + decode + LANMS", "batch": n, "size": S, "steps": 3,
                          "ms_per_step": round((dt_net + dt_dec + dt_nms) * 1e3, 3),
                          "images_per_sec": round(n / (dt_net + dt_dec + dt_nms), 1),
                          "tflops": round(fwd_tf, 1), "frac_of_peak": round(fwd_tf / MFMA_PEAK_TFLOPS, 4),
                          "dominant_kernel": dominant_kernel(timing),
                          "net_forward_ms": round(dt_net * 1e3, 2), "net_images_per_sec": round(n / dt_net, 1),
                          "decode_ms_per_batch": round(dt_dec * 1e3, 3), "decode_images_per_sec": round(n / dt_dec, 1),
                          "decode_algorithmic_GBps": round(px * 76 / dt_dec / 1e9, 1), "components": ncomp,
                          "lanms_ms_per_batch": round(dt_nms * 1e3, 3),
                          "lanms_boxes_per_sec": round(n * K / dt_nms, 0)}), flush=True)


    if "pipeline" in which:
        pipeline_config()
    if "east_fwd" in which:
        east_fwd_config()
    if "f32_forward" in which:
        f32_forward_config(args.steps, args.warmup)


def east_fwd_config():
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import model
    from tensorflow_ocr_amd.tool import pixellink_fn
    dev = torch.device("cuda", 0)
    g = Graph(dev, seed=1)
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.uniform(0, 255, (1, 512, 512, 3)).astype(np.float32)).to(dev)
    holder = {}

    def net():
        a, b = model.model(x, is_training=False, graph=g)
        g.reset_tape()
        holder["o"] = (a, b)
    net()
    dt_net = timed(net, 2, 10)
    from tensorflow_ocr_amd.infer import GraphedForward

    def fwd_fn(gr, im):
        a, b = model.model(im, is_training=False, graph=gr)
        return a, b
    graphed = GraphedForward(g, fwd_fn)
    graphed(x)
    dt_graph = timed(lambda: graphed(x), 2, 20)
    # decode on a synthetic, well-behaved score map (random-init weights give degenerate maps)
    sc = torch.from_numpy((rng.uniform(size=(1, 128, 128, 1)) < 0.12).astype(np.float32)).to(dev)
    ys, xs = np.mgrid[0:128, 0:128]
    blob = np.zeros((128, 128), np.float32)
    for _ in range(12):
        cy, cx = rng.uniform(8, 120, 2)
        blob[(np.abs(xs - cx) < rng.uniform(4, 20)) & (np.abs(ys - cy) < rng.uniform(2, 6))] = 1
    sc = torch.from_numpy(blob[None, :, :, None]).to(dev)
    geo = torch.from_numpy(rng.uniform(0.5, 1.0, (1, 128, 128, 16)).astype(np.float32)).to(dev)

    def dec():
        m = pixellink_fn.east_pixel_detect(sc, geo, 0.8, 0.8, graph=g)
        holder["b"] = pixellink_fn.find_contour_boxes(m, graph=g)
    dec()
    dt_dec = timed(dec, 1, 5)
    # CPU restatement of the same forward (f32, torch-CPU): the reference's own configuration for this config
    import os
    from oracle import ocr_oracle as O
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    p = O.init_model_resnet_params(np.random.default_rng(1)) if hasattr(O, "init_model_resnet_params") else None
    cpu = None
    if p is not None:
        tp = O.to_torch_params(p)
        xi = x.cpu()
        with torch.no_grad():
            O.model_resnet(xi, tp, False)
            t0 = time.perf_counter()
            O.model_resnet(xi, tp, False)
            cpu = time.perf_counter() - t0
    print(json.dumps({"config": "test.py path, one 512x512 image: model.model inference + EAST-script decode (configs[0])",
                      "net_ms_hip_graph": round(dt_graph * 1e3, 3),
                      "net_forward_ms": round(dt_net * 1e3, 3), "images_per_sec": round(1 / dt_net, 1),
                      "decode_ms": round(dt_dec * 1e3, 3), "boxes": int(len(holder["b"][1])),
                      "cpu_baseline": None if cpu is None else {"kind": "port", "cores": torch.get_num_threads(),
                                                                  "forward_ms": round(cpu * 1e3, 1),
                                                                  "images_per_sec": round(1 / cpu, 2)}}), flush=True)


def f32_forward_config(steps, warmup):
    """The precision that MEETS the north star's "score maps within 1e-3 of the f32 reference" (missing 5):
    Graph(precision="f32") — f32 storage, every convolution on the matrix cores with v_mfma_f32_32x32x2_f32 (exact f32
    products and sums; test.py --precision f32, tests/test_gpu_product_accuracy.py: L-inf 2e-5 .. 2e-4) — timed as the
    inference forward it is used for: model_vgg at 512^2 (configs[1]'s graph, batch 8) and PixelLinkNet at 1024^2
    (configs[4]'s graph, batch 2).  The f16 training path is the headline; this leg puts a driver-timed number on the
    accurate one.  Peak for this dtype: 157 TFLOP/s (MI355X_MICROARCH.md: f32 MFMA = 1/16 of the 16-bit rate)."""
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import model_vgg_16 as M
    from tensorflow_ocr_amd.nets import pixellink
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    out = {}
    for key, size, batch, gflop in (("model_vgg_512", 512, 8, 172.5), ("pixellinknet_1024", 1024, 2, 689.8)):
        g = Graph(dev, seed=1, precision="f32")
        x = torch.from_numpy(rng.uniform(0, 255, (batch, size, size, 3)).astype(np.float32)).to(dev)

        def fwd():
            if key.startswith("model_vgg"):
                M.model_vgg(x, is_training=False, graph=g)
            else:
                pixellink.PixelLinkNet(x, graph=g)
            g.reset_tape()
        fwd()
        dt = timed(fwd, warmup, steps)
        out[key] = {"batch": batch, "size": size, "ms_per_forward": round(dt * 1e3, 2), "images_per_sec": round(batch / dt, 1),
                    "tflops": round(gflop * batch / dt / 1e3, 1), "frac_of_f32_mfma_peak": round(gflop * batch / dt / 1e3 / 157.3, 3)}
        del g
        torch.cuda.empty_cache()
    a = out["model_vgg_512"]
    print(json.dumps({"config": "f32 inference forward (Graph(precision='f32'), v_mfma_f32_32x32x2_f32): model_vgg 512^2 b8, "
                                "PixelLinkNet 1024^2 b2 — the path that meets the 1e-3 score-map bar",
                      "dtype": "f32", "steps": steps, "ms_per_step": a["ms_per_forward"], "images_per_sec": a["images_per_sec"],
                      "tflops": a["tflops"], "frac_of_peak": a["frac_of_f32_mfma_peak"], "components": out}), flush=True)


def dev_ms(fn, warmup=2, steps=10):
    """Device time of fn() from HIP events on the current stream."""
    for _ in range(warmup):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def pipeline_config():
    from tensorflow_ocr_amd import ops
    from tensorflow_ocr_amd.datasets import icdar
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.tool import pixellink_fn
    dev = torch.device("cuda", 0)
    g = Graph(dev)
    rng = np.random.default_rng(9)
    n, S, K = 32, 512, 16

    def quads(k, size):
        out = []
        for _ in range(k):
            c = rng.uniform(20, size - 20, 2)
            w, h = rng.uniform(30, size / 3), rng.uniform(10, size / 10)
            th = rng.uniform(-0.5, 0.5)
            R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
            out.append(np.clip((np.array([[-w, -h], [w, -h], [w, h], [-w, h]]) / 2) @ R.T + c, 0, size - 1))
        return np.array(out, np.float32)
    polys_l = [quads(K, S) for _ in range(n)]
    tags_l = [rng.uniform(size=K) < 0.2 for _ in range(n)]
    # --- label maps
    polys, counts, ignore = icdar.pack_polys(polys_l, tags_l)
    d_p, d_c, d_i = [torch.from_numpy(a).to(dev) for a in (polys, counts, ignore)]
    cover = torch.empty((n, S, S), dtype=torch.int32, device=dev)
    score = torch.empty((n, S // 4, S // 4, 1), device=dev)
    geo = torch.empty((n, S // 4, S // 4, 8), device=dev)
    mask = torch.empty((n, S // 4, S // 4, 1), device=dev)
    ms_cover = dev_ms(lambda: ops.poly_cover(d_p, d_c, d_i, S, S, cover))
    ms_lab = dev_ms(lambda: ops.icdar_labels(cover, 4, score, geo, mask))
    t_e2e = timed(lambda: icdar.generate_rbox_batch((S, S), polys_l, tags_l, graph=g), 2, 10)
    lab_bytes = n * (S * S * 4 + (S // 4) ** 2 * (9 * 4 + 40))       # cover write + 9 neighbour reads + 10 f32 out
    # --- resize 720x1280 -> 512x512
    ims = [rng.integers(0, 256, size=(720, 1280, 3)).astype(np.uint8) for _ in range(n)]
    src = torch.from_numpy(ims[0]).to(dev)
    dst = torch.empty((S, S, 3), device=dev)
    ms_resize = dev_ms(lambda: ops.resize_linear_u8(src, dst)) * n
    t_resize_e2e = timed(lambda: icdar.resize_images(ims, S, graph=g), 1, 5)
    # --- oriented boxes of decoded components (16 x 256^2 maps with rotated text-like blobs)
    nb, hq = 16, 256
    lab = np.zeros((nb, hq, hq), np.int32)
    ncomp = np.zeros(nb, np.int32)
    ys, xs = np.mgrid[0:hq, 0:hq]
    for b in range(nb):
        k = 0
        for _ in range(24):
            cy, cx = rng.uniform(10, hq - 10, 2)
            a, bb, th = rng.uniform(8, 40), rng.uniform(2, 7), rng.uniform(-0.6, 0.6)
            u = (xs - cx) * np.cos(th) + (ys - cy) * np.sin(th)
            v = -(xs - cx) * np.sin(th) + (ys - cy) * np.cos(th)
            m = (np.abs(u) <= a) & (np.abs(v) <= bb) & (lab[b] == 0)
            if m.sum() > 10:
                k += 1
                lab[b][m] = k
        ncomp[b] = k
    d_lab, d_nc = torch.from_numpy(lab).to(dev), torch.from_numpy(ncomp).to(dev)
    hn = torch.zeros((nb, 64), dtype=torch.int32, device=dev)
    hd = torch.zeros((nb, 64, 4), dtype=torch.int32, device=dev)
    cal = torch.zeros((nb, 64, 6), device=dev)
    ms_mar = dev_ms(lambda: ops.min_area_rects(d_lab, d_nc, 64, 4.0, 4.0, hn, hd, cal, g.workspace()))
    t_boxes_e2e = timed(lambda: pixellink_fn.min_area_rect_boxes(d_lab, d_nc, 4.0, 4.0, max_comps=64, graph=g), 1, 5)
    # --- CPU restatement beside it (bounded samples; baseline only)
    from oracle import cvgeom as C
    from oracle import labels as OL
    t0 = time.perf_counter()
    OL.icdar_labels((S, S), polys_l[0], tags_l[0])
    cpu_lab = time.perf_counter() - t0
    t0 = time.perf_counter()
    C.resize_linear_u8(ims[0], S, S)
    cpu_resize = time.perf_counter() - t0
    t0 = time.perf_counter()
    for i in range(1, ncomp[0] + 1):
        xy = np.argwhere(lab[0] == i)
        show = xy.copy()
        show[:, 0] = xy[:, 1] * 4.0
        show[:, 1] = xy[:, 0] * 4.0
        C.box_points(C.min_area_rect(show)[0])
    cpu_boxes = time.perf_counter() - t0
    print(json.dumps({
        "config": "input pipeline + boxes (SURVEY 8f): 32 x 512^2 labels (16 quads), 32 x 720x1280 -> 512^2 resize, 16 x 256^2 boxes",
        "labels_cover_ms": round(ms_cover, 4), "labels_maps_ms": round(ms_lab, 4),
        "labels_GBps": round(lab_bytes / ((ms_cover + ms_lab) * 1e-3) / 1e9, 1),
        "labels_images_per_sec_kernels": round(n / ((ms_cover + ms_lab) * 1e-3), 0),
        "labels_e2e_ms_incl_host_pack": round(t_e2e * 1e3, 3),
        "resize_ms_per_32": round(ms_resize, 4),
        "resize_GBps": round(n * (720 * 1280 * 3 + S * S * 12) / (ms_resize * 1e-3) / 1e9, 1),
        "resize_e2e_ms_incl_pcie": round(t_resize_e2e * 1e3, 3),
        "resize_e2e_pcie_GBps": round(n * 720 * 1280 * 3 / t_resize_e2e / 1e9, 1),
        "boxes_kernels_ms": round(ms_mar, 4), "boxes_components": int(ncomp.sum()),
        "boxes_e2e_ms_incl_host_format": round(t_boxes_e2e * 1e3, 3),
        "cpu_baseline": {"kind": "port", "cores": 1,
                         "labels_images_per_sec": round(1 / cpu_lab, 2), "resize_images_per_sec": round(1 / cpu_resize, 1),
                         "boxes_components_per_sec": round(int(ncomp[0]) / cpu_boxes, 0),
                         "sample": "one 512^2 image of labels (python loops over the C fillPoly), one 720x1280 resize (C), one 256^2 map of boxes (C)"}}),
        flush=True)


if __name__ == "__main__":
    main()
