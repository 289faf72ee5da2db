#!/usr/bin/env python3
"""Counterpart of the reference's test_pixellink.py (eval :44-232): the FULL-RESOLUTION decode.  Same
network and softmaxes as test_pixellink_fast.py; then the nine score maps are up-sampled to 1280x720
with cv2.resize(INTER_CUBIC) (:97-98, :108-109), pixels are kept where `pixel_score * 255 > 204`,
links where `link * 255 > 229.5` (:111,:119), groups need more than 200 pixels (:177), and each
group's box is `np.int0(cv2.boxPoints(cv2.minAreaRect(xy)))` in full-resolution coordinates
(:208-218), written as `res_<name>.txt` (:223-232).

Flag names follow the reference (:12-19).  The up-sampling, the grouping and the hull + calipers run
on the GPU (tool/pixellink_fn.full_resolution_decode, min_area_rect_boxes).  Forced differences
(SURVEY D5/§8c): .npy images only, no annotated jpgs."""
import argparse
import os
import time

import numpy as np
import torch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--checkpoint_path', type=str, default='./ohem_logs/')
    ap.add_argument('--test_data_path', type=str, default='/tmp/images/')
    ap.add_argument('--output_dir', type=str, default='/tmp/output/')
    ap.add_argument('--gpu_memory_fraction', type=float, default=-1)
    ap.add_argument('--pixel_conf_threshold', type=float, default=0.8)
    ap.add_argument('--link_conf_threshold', type=float, default=0.8)
    ap.add_argument('--eval_image_width', type=int, default=1280)
    ap.add_argument('--eval_image_height', type=int, default=768)
    ap.add_argument('--decode_width', type=int, default=1280, help='cv2.resize target (:98), width')
    ap.add_argument('--decode_height', type=int, default=720, help='cv2.resize target (:98), height')
    ap.add_argument('--synthetic', type=int, default=0, help='decode N synthetic images instead of files')
    ap.add_argument('--precision', choices=['f16', 'f32'], default='f16', help="f32: the forward pass in the f32 inference "
                    "precision (f32 storage, matrix-core f32 convolutions): score maps within 1e-3 of the f32 reference, ~10x the time")
    return ap.parse_args()


def get_images(path):
    files = []
    for parent, _, filenames in os.walk(path):
        for f in sorted(filenames):
            if f.endswith('.npy'):
                files.append(os.path.join(parent, f))
    print('Find {} images'.format(len(files)))
    return files


def main():
    FLAGS = parse()
    from tensorflow_ocr_amd import checkpoint
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.infer import GraphedForward
    from tensorflow_ocr_amd.nets import pixellink
    from tensorflow_ocr_amd.tool import pixellink_fn
    g = Graph('cuda:0', precision=FLAGS.precision)
    os.makedirs(FLAGS.output_dir, exist_ok=True)
    H, W = FLAGS.eval_image_height, FLAGS.eval_image_width
    if FLAGS.synthetic:
        rng = np.random.default_rng(0)
        items = [('synthetic_%d' % i, rng.uniform(0, 255, (H, W, 3)).astype(np.float32)) for i in range(FLAGS.synthetic)]
    else:
        items = [(os.path.basename(f).split('.')[0], np.load(f).astype(np.float32)) for f in get_images(FLAGS.test_data_path)]
    loaded = False

    def network(gr, im):      # sess.run of the pixel / link softmaxes: one HIP graph (every image has the eval size)
        net = pixellink.PixelLinkNet(im, graph=gr)
        return net.pixel_scores, pixellink_fn.link_scores(net.link_cls, graph=gr)
    forward = GraphedForward(g, network)
    for name, im in items:
        if im.shape[:2] != (H, W):
            raise SystemExit('%s: expected %dx%d input' % (name, H, W))
        x = torch.from_numpy(((im - 120.0) / 60.0)[None]).to(g.device)
        t0 = time.time()
        pixel_score, link_score = forward(x)                            # [1,h,w,2], [8,1,h,w,2]
        if FLAGS.checkpoint_path and not loaded:
            src = FLAGS.checkpoint_path
            if src.endswith('.npz'):
                sd = dict(np.load(src))
            elif os.path.exists(os.path.join(src, 'checkpoint')) or os.path.exists(src + '.index') or os.path.isfile(src):
                sd, _ = checkpoint.load_tf_checkpoint(src, use_moving_averages=True)    # get_checkpoint_state + restore (:76-84)
            else:
                sd = None
            if sd is not None:
                g.store.load_state_dict(checkpoint.tf_to_internal(g.store.order, sd), strict=False)
                pixel_score, link_score = forward(x)
            loaded = True
        labels, ncomp, _ = pixellink_fn.full_resolution_decode(
            pixel_score[..., 1].contiguous(), link_score, FLAGS.decode_height, FLAGS.decode_width, graph=g)
        k = int(ncomp[0].item())
        print('%s: net+decode %.0f ms, %d groups' % (name, (time.time() - t0) * 1e3, k))
        # show_xy = (x, y) of the group's pixels; rectangle = cv2.minAreaRect(show_xy)  (:208-216)
        _, boxes = pixellink_fn.min_area_rect_boxes(labels, ncomp, 1.0, 1.0, graph=g)[0]
        with open(os.path.join(FLAGS.output_dir, 'res_{}.txt'.format(name)), 'w') as f:
            for box in boxes:
                f.write('{},{},{},{},{},{},{},{}\r\n'.format(box[0, 0], box[0, 1], box[1, 0], box[1, 1],
                                                            box[2, 0], box[2, 1], box[3, 0], box[3, 1]))


if __name__ == '__main__':
    main()
