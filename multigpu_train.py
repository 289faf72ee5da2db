#!/usr/bin/env python3
"""Counterpart of the reference's multigpu_train.py (flags :6-17, tower/step assembly :27-142, hot
loop :169-194) on the MI355X path: one process per GPU

    python multigpu_train.py --gpu_list 0 --batch_size_per_gpu 14 --input_size 512
    python multigpu_train.py --gpu_list 0,1           (starts one rank per listed GPU itself)
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 multigpu_train.py --gpu_list 0,1

Same flag names and defaults; `--net` selects the graph (`model` = nets/model.py ResNet-50 +
PixelLink heads + OHNM loss, the one the reference script imports; `model_vgg` / `east` /
`pixellink` the others).  Data: `--training_data_path` (the reference's icdar.py flag) feeds ICDAR
images + gt_*.txt through datasets/icdar.get_batch — host decode in `--num_readers` processes,
upload + resize + label maps on the feeder's HIP stream, overlapped with the step; without it (or
with an empty directory) batches are synthetic (`tensorflow_ocr_amd.synthetic`).  The stdout line
format is the reference's (:183-184)."""
import argparse
import os
import time

import numpy as np
import torch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--input_size', type=int, default=512)
    ap.add_argument('--batch_size_per_gpu', type=int, default=14)
    ap.add_argument('--num_readers', type=int, default=16)
    ap.add_argument('--learning_rate', type=float, default=0.0001)
    ap.add_argument('--max_steps', type=int, default=100000)
    ap.add_argument('--moving_average_decay', type=float, default=0.997)
    ap.add_argument('--gpu_list', type=str, default='1')
    ap.add_argument('--checkpoint_path', type=str, default='/tmp/east_resnet_v1_50_rbox/')
    ap.add_argument('--restore', action='store_true')
    ap.add_argument('--save_checkpoint_steps', type=int, default=1000)
    ap.add_argument('--save_summary_steps', type=int, default=20)
    ap.add_argument('--pretrained_model_path', type=str, default=None)
    ap.add_argument('--training_data_path', type=str, default='/data/ocr/icdar2015/')
    ap.add_argument('--net', choices=['model', 'model_vgg', 'east', 'pixellink'], default='model')
    return ap.parse_args()


def build_forward_loss(net):
    from tensorflow_ocr_amd.nets import model, model_vgg_16, pixellink
    if net == 'model':
        def f(g, im, sm, gm, tm):
            a, b = model.model(im, is_training=True, graph=g)
            return model.loss(sm, a, gm, b, tm, graph=g)
    elif net == 'model_vgg':
        def f(g, im, sm, gm, tm):
            a, b = model_vgg_16.model_vgg(im, is_training=True, graph=g)
            return model_vgg_16.loss(sm, a, gm, b, tm, graph=g)
    elif net == 'east':
        def f(g, im, sm, gm, tm):
            a, b = model_vgg_16.model(im, is_training=True, graph=g)
            return model_vgg_16.loss(sm, a, gm, b, tm, graph=g)
    else:
        def f(g, im, sm, gm, tm):
            # raw images in; the pipeline's (x - 120) / 60 runs inside the image-preparation kernel
            n = pixellink.PixelLinkNet(im, graph=g, input_norm=(120.0, 60.0))
            return n.build_loss(sm[..., 0], gm)
    return f


def main():
    FLAGS = parse()
    gpus = FLAGS.gpu_list.split(',')
    # os.environ['CUDA_VISIBLE_DEVICES'] = FLAGS.gpu_list (multigpu_train.py:90) + one tower per entry
    # (:118-130).  Started plainly, this process either IS the single tower or becomes the launcher of
    # len(gpus) ranks; nothing has touched HIP yet.
    from tensorflow_ocr_amd import launch
    if 'WORLD_SIZE' not in os.environ:
        if len(gpus) > 1:
            raise SystemExit(launch.self_launch(len(gpus), visible=FLAGS.gpu_list))
        os.environ.setdefault('HIP_VISIBLE_DEVICES', FLAGS.gpu_list)
    from tensorflow_ocr_amd import checkpoint, dist, synthetic
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.train import AdamOptimizer, TrainStep
    rank, world, local = dist.init_process_group_from_env()
    if world not in (1, len(gpus)):
        raise SystemExit("gpu_list has %d entries but WORLD_SIZE=%d" % (len(gpus), world))
    device = torch.device('cuda', local)     # the launcher maps LOCAL_RANK -> visible GPU
    torch.cuda.set_device(device)
    if rank == 0:
        os.makedirs(FLAGS.checkpoint_path, exist_ok=True)

    g = Graph(device, seed=1)
    step = TrainStep(g, build_forward_loss(FLAGS.net),
                     lambda gr: AdamOptimizer(gr, learning_rate=FLAGS.learning_rate,
                                              moving_average_decay=FLAGS.moving_average_decay),
                     world_size=world)
    rng = np.random.default_rng(1000 + rank)
    feeder = None
    if os.path.isdir(FLAGS.training_data_path):
        from tensorflow_ocr_amd.datasets import icdar
        if icdar.get_images(FLAGS.training_data_path):
            # multigpu_train.py:164-167: icdar.get_batch(num_workers, input_size, batch_size)
            feeder = icdar.get_batch(num_workers=FLAGS.num_readers, training_data_path=FLAGS.training_data_path,
                                     input_size=FLAGS.input_size, batch_size=FLAGS.batch_size_per_gpu,
                                     graph=g, seed=1000 + rank)
    start = time.time()
    try:
        _train_loop(FLAGS, g, step, feeder, rng, rank, world, device, start)
    finally:
        if feeder is not None:
            feeder.close()


def _next_batch(FLAGS, feeder, rng, device):
    from tensorflow_ocr_amd import synthetic
    if feeder is not None:
        images, _, score_maps, geo_maps, training_masks = next(feeder)
        return [images, score_maps, geo_maps, training_masks]
    data = synthetic.make_batch(rng, FLAGS.batch_size_per_gpu, FLAGS.input_size)
    return [torch.from_numpy(a).to(device, non_blocking=True) for a in data]


def _train_loop(FLAGS, g, step, feeder, rng, rank, world, device, start):
    from tensorflow_ocr_amd import checkpoint, dist
    batch = _next_batch(FLAGS, feeder, rng, device)
    # variables, optimiser slots and EMA shadows exist before the first update, so a checkpoint is
    # restored (:153-158) / the pretrained backbone loaded (:149-151,161-162) BEFORE it, as in the reference
    step.build(*batch)
    opt = step.opt
    src = FLAGS.checkpoint_path if FLAGS.restore else FLAGS.pretrained_model_path
    if src:
        if not (os.path.isdir(src) and os.path.exists(os.path.join(src, 'checkpoint')) or os.path.exists(src + '.index')
                or os.path.isfile(src)):
            raise FileNotFoundError('no checkpoint at %s' % src)
        if FLAGS.restore:
            checkpoint.restore_training_state(src, g, opt)
            if opt.restored_variables == 0:
                raise ValueError('no variable of this graph found in %s' % src)
            if rank == 0:
                print('continue training from previous checkpoint (%d variables)' % opt.restored_variables)
        else:
            sd, _ = checkpoint.load_tf_checkpoint(src)        # variables only (slim.assign_from_checkpoint_fn)
            g.store.load_state_dict(checkpoint.tf_to_internal(g.store.order, sd), strict=False)
            if opt.ema is not None:
                opt.ema.copy_(g.store.flat)
            if rank == 0:
                print('loaded ' + src)
    # `for step in range(FLAGS.max_steps)` (multigpu_train.py:168): the loop counter is separate from the restored
    # `global_step` (which the optimiser continues from), so a resumed run takes max_steps MORE steps
    for it in range(FLAGS.max_steps):
        if it > 0:
            batch = _next_batch(FLAGS, feeder, rng, device)
        loss = step(*batch)
        if it % 10 == 0:
            ml = loss.item()
            # the stop decision is collective: a rank leaving alone would strand the others in the
            # next gradient all-reduce
            if dist.any_rank(bool(np.isnan(ml))):
                if rank == 0:
                    print('Loss diverged, stop training')
                break
            avg_time_per_step = (time.time() - start) / 10
            avg_examples_per_second = (10 * FLAGS.batch_size_per_gpu * world) / (time.time() - start)
            start = time.time()
            if rank == 0:
                # total_loss = model_loss + sum(REGULARIZATION_LOSSES) (multigpu_train.py:36)
                tl = ml + opt.regularization_loss().item()
                print('Step {:06d}, model loss {:.4f}, total loss {:.4f}, {:.2f} seconds/step, {:.2f} examples/second'.format(
                    it, ml, tl, avg_time_per_step, avg_examples_per_second), flush=True)
        if rank == 0 and it % FLAGS.save_checkpoint_steps == 0:     # step 0 included (multigpu_train.py:185-186)
            # saver.save(sess, FLAGS.checkpoint_path + 'model.ckpt', global_step=global_step) (:186-187):
            # a TensorFlow V2 bundle of Saver(tf.global_variables()) — variables, EMA shadows, Adam slots
            checkpoint.save_training_state(FLAGS.checkpoint_path, g, opt)


if __name__ == '__main__':
    main()
