#!/usr/bin/env python3
"""Counterpart of the reference's test_pixellink_fast.py (eval :44-217): PixelLinkNet forward,
per-direction softmax, pixel_detect mask, link-gated connected components at 1/4 resolution, one
box per component, `res_<name>.txt` lines `x1,y1,x2,y2,x3,y3,x4,y4\\r\\n` (:215-217).

Flag names follow the reference (:12-19).  Differences, all forced by the container:
images are read with NumPy-decodable formats only (.npy arrays [H,W,3] RGB; no cv2).  The box of a
component is `np.int0(cv2.boxPoints(cv2.minAreaRect(show_xy)))` as in the reference (:193-202), with
hull + rotating calipers on the GPU (tool/pixellink_fn.min_area_rect_boxes)."""
import argparse
import os
import time

import numpy as np
import torch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--checkpoint_path', type=str, default=None)
    ap.add_argument('--test_data_path', type=str, default='/tmp/images/')
    ap.add_argument('--output_dir', type=str, default='/tmp/output/')
    ap.add_argument('--gpu_memory_fraction', type=float, default=-1)
    ap.add_argument('--pixel_conf_threshold', type=float, default=0.8)
    ap.add_argument('--link_conf_threshold', type=float, default=0.9)
    ap.add_argument('--eval_image_width', type=int, default=1280)
    ap.add_argument('--eval_image_height', type=int, default=768)
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
            raise SystemExit('%s: expected %dx%d input (resize is a cv2 step, out of scope)' % (name, H, W))
        x = torch.from_numpy(((im - 120.0) / 60.0)[None]).to(g.device)
        t0 = time.time()
        pixel_score, link_score = forward(x)                            # [1,h,w,2], [8,1,h,w,2]
        if FLAGS.checkpoint_path and not loaded:
            if FLAGS.checkpoint_path.endswith('.npz'):
                sd = dict(np.load(FLAGS.checkpoint_path))
            else:       # a TF checkpoint directory / prefix; EMA shadows restored like test.py:149-150
                sd, _ = checkpoint.load_tf_checkpoint(FLAGS.checkpoint_path, use_moving_averages=True)
            g.store.load_state_dict(checkpoint.tf_to_internal(g.store.order, sd), strict=False)
            loaded = True
            pixel_score, link_score = forward(x)
        score_res = pixellink_fn.tf_pixel_detect(pixel_score[..., 1:2].contiguous(), link_score,
                                                 FLAGS.pixel_conf_threshold, FLAGS.link_conf_threshold, graph=g)
        labels, ncomp, comps = pixellink_fn.link_cc_decode(pixel_score[..., 1].contiguous(), link_score,
                                                           FLAGS.pixel_conf_threshold, FLAGS.link_conf_threshold,
                                                           min_size=10, graph=g)
        k = int(ncomp[0].item())
        print('%s: net+decode %.0f ms, %d components, mask pixels %d' % (
            name, (time.time() - t0) * 1e3, k, int(score_res.sum().item())))
        h4, w4 = labels.shape[1:]
        # rectangle = cv2.minAreaRect(show_xy); box = np.int0(cv2.boxPoints(rectangle))  (:193-202)
        _, boxes = pixellink_fn.min_area_rect_boxes(labels, ncomp, float(W) / w4, float(H) / h4, graph=g)[0]
        with open(os.path.join(FLAGS.output_dir, 'res_{}.txt'.format(name)), 'w') as f:
            for box in boxes:
                f.write('{},{},{},{},{},{},{},{}\r\n'.format(box[0, 0], box[0, 1], box[1, 0], box[1, 1],
                                                            box[2, 0], box[2, 1], box[3, 0], box[3, 1]))


if __name__ == '__main__':
    main()
