#!/usr/bin/env python3
"""Counterpart of the reference's train_pixellink.py (flags :17-79, clone/optimiser assembly
:196-283) on the MI355X path: one process per GPU ("clone")

    python train_pixellink.py --dataset_dir /data/icdar2015/train --batch_size 32 --num_gpus 1
    python train_pixellink.py --num_gpus 2 ...        (starts the two clones itself)
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 train_pixellink.py --num_gpus 2 ...

Same flag names and defaults where they bear on the step: `PixelLinkNet` + `build_loss`
(nets/pixellink.py), each clone's loss divided by num_clones (:264) and the gradients SUMMED
(`sum_gradients`, :179-194: `TrainStep(grad_op="sum")` seeds the backward pass with 1/num_clones and
all-reduces with SUM), Momentum 0.9,
`tf.case` staircase on --lr_breakpoints / --lr_decays (:222-237), weight decay 5e-4.

Data: the reference reads TFRecords through `datasets.dataset_factory` / `ssd_vgg_preprocessing`,
neither of which is in its tree (not runnable as shipped).  Here --dataset_dir is an
ICDAR directory (images + gt_*.txt); every image is resized to the train size on the GPU and its
quads go, normalised, through `pixellink_fn.generate_rbox` (tool/pixellink_fn.py:53-110) — the label
generator this script's `tf_pixellink_get_rbox` wraps.  Without --dataset_dir: synthetic batches."""
import argparse
import os
import time

import numpy as np
import torch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--train_with_ignored', action='store_true')
    ap.add_argument('--train_dir', type=str, default=None)
    ap.add_argument('--checkpoint_path', type=str, default=None)
    ap.add_argument('--batch_size', type=int, default=32, help='global batch over all clones')
    ap.add_argument('--num_gpus', type=int, default=1)
    ap.add_argument('--max_number_of_steps', type=int, default=60000)
    ap.add_argument('--log_every_n_steps', type=int, default=10)
    ap.add_argument('--learning_rate', type=float, default=0.01)
    ap.add_argument('--lr_policy', type=str, default='staircase')
    ap.add_argument('--lr_breakpoints', type=str, default='20000,40000,60000')
    ap.add_argument('--lr_decays', type=str, default='0.1,0.01,0.001')
    ap.add_argument('--momentum', type=float, default=0.9)
    ap.add_argument('--weight_decay', type=float, default=0.0005)
    ap.add_argument('--using_moving_average', action='store_true')
    ap.add_argument('--moving_average_decay', type=float, default=0.9999)
    ap.add_argument('--num_readers', type=int, default=32)
    ap.add_argument('--dataset_dir', type=str, default=None)
    ap.add_argument('--train_image_width', type=int, default=512)
    ap.add_argument('--train_image_height', type=int, default=512)
    return ap.parse_args()


def staircase_lr(step, base, breakpoints, decays):
    """train_pixellink.py:222-237: tf.case over (step < breakpoint_i -> decay_i), default 1.0."""
    for bp, dc in zip(breakpoints, decays):
        if step < bp:
            return base * dc
    return base * 1.0


# The input normalisation of the reference's pipeline (train_pixellink.py:150-154 hands the queue
# `ssd_vgg_preprocessing` output): (x - 120) / 60, applied by the net's image-preparation kernel
INPUT_NORM = (120.0, 60.0)


def dataset_batches(FLAGS, g, batch, rank):
    """ICDAR directory -> (images [B,H,W,3] f32, pixel labels [B,H/4,W/4], link labels [B,H/4,W/4,8])."""
    from tensorflow_ocr_amd.datasets import icdar
    from tensorflow_ocr_amd.tool import pixellink_fn
    H, W = FLAGS.train_image_height, FLAGS.train_image_width
    files = sorted(icdar.get_images(FLAGS.dataset_dir))
    rng = np.random.RandomState(1000 + rank)
    while True:
        rng.shuffle(files)
        ims, xs_l, ys_l, bb_l, ig_l = [], [], [], [], []
        for fn in files:
            tf = icdar.txt_name(fn)
            if not os.path.exists(tf):
                continue
            im = icdar.read_image_rgb(fn)
            h, w, _ = im.shape
            polys, tags = icdar.load_annoataion(tf)
            polys, tags = icdar.check_and_validate_polys(polys, tags, (h, w))
            if len(polys) == 0:
                continue
            xs, ys = polys[:, :, 0] / w, polys[:, :, 1] / h
            ims.append(im)
            xs_l.append(xs)
            ys_l.append(ys)
            bb_l.append(np.stack([ys.min(1), xs.min(1), ys.max(1), xs.max(1)], 1))
            ig_l.append(tags.astype(np.int32))
            if len(ims) == batch:
                if H == W:
                    images = icdar.resize_images(ims, H, graph=g)
                else:
                    raise SystemExit('square train size only (resize_images)')
                score, link, _ = pixellink_fn.generate_rbox_batch(H, W, xs_l, ys_l, bb_l, ig_l, graph=g)
                yield images, score, link
                ims, xs_l, ys_l, bb_l, ig_l = [], [], [], [], []


def main():
    FLAGS = parse()
    from tensorflow_ocr_amd import launch
    rc = launch.self_launch(FLAGS.num_gpus)          # plain start with --num_gpus N: become the launcher
    if rc is not None:
        raise SystemExit(rc)
    from tensorflow_ocr_amd import checkpoint, dist, synthetic
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import pixellink
    from tensorflow_ocr_amd.train import MomentumOptimizer, TrainStep
    rank, world, local = dist.init_process_group_from_env()
    if world != FLAGS.num_gpus:
        raise SystemExit('--num_gpus %d but WORLD_SIZE=%d' % (FLAGS.num_gpus, world))
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    batch_size_per_gpu = int(FLAGS.batch_size / FLAGS.num_gpus)          # :92
    bps = [int(o) for o in FLAGS.lr_breakpoints.split(',')]
    dcs = [float(o) for o in FLAGS.lr_decays.split(',')]
    assert len(bps) == len(dcs)
    if FLAGS.lr_policy != 'staircase':
        raise SystemExit('Unkonw lr_policy: {}'.format(FLAGS.lr_policy))

    g = Graph(device, seed=1)

    def forward_loss(gr, im, pixel_labels, link_labels):
        net = pixellink.PixelLinkNet(im, graph=gr, input_norm=INPUT_NORM)   # raw images in
        return net.build_loss(pixel_labels, link_labels)                 # 2*pixel + link (two LOSSES, :263)

    def make_opt(gr):
        opt = MomentumOptimizer(gr, base_lr=FLAGS.learning_rate, momentum=FLAGS.momentum,
                                weight_decay=FLAGS.weight_decay,
                                moving_average_decay=FLAGS.moving_average_decay if FLAGS.using_moving_average else None)
        opt.learning_rate = lambda: staircase_lr(opt.global_step, FLAGS.learning_rate, bps, dcs)
        return opt
    step = TrainStep(g, forward_loss, make_opt, world_size=world, grad_op="sum")

    feeder = None
    if FLAGS.dataset_dir and os.path.isdir(FLAGS.dataset_dir):
        from tensorflow_ocr_amd.feeder import DeviceFeeder
        feeder = DeviceFeeder(lambda: dataset_batches(FLAGS, g, batch_size_per_gpu, rank), device, depth=2)
    rng = np.random.default_rng(1000 + rank)
    if FLAGS.train_dir and rank == 0:
        os.makedirs(FLAGS.train_dir, exist_ok=True)
    start = time.time()
    for it in range(FLAGS.max_number_of_steps):
        if feeder is not None:
            images, pixel, link = next(feeder)
        else:
            im, px, lk, _ = synthetic.make_batch(rng, batch_size_per_gpu, FLAGS.train_image_height)
            images, pixel, link = [torch.from_numpy(a).to(device, non_blocking=True) for a in (im, px[..., 0], lk)]
        loss = step(images, pixel, link)
        if it % FLAGS.log_every_n_steps == 0:
            v = loss.item()
            dt = (time.time() - start) / FLAGS.log_every_n_steps
            start = time.time()
            if rank == 0:
                print('global step %d: loss = %.4f (%.3f sec/step), lr %.6f' % (
                    it, v, dt, staircase_lr(it, FLAGS.learning_rate, bps, dcs)), flush=True)
            if dist.any_rank(bool(np.isnan(v))):     # collective: no rank leaves the all-reduce alone
                break
        if FLAGS.train_dir and rank == 0 and it > 0 and it % 1000 == 0:
            checkpoint.save_training_state(FLAGS.train_dir, g, step.opt)
    if feeder is not None:
        feeder.close()


if __name__ == '__main__':
    main()
