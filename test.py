#!/usr/bin/env python3
"""Counterpart of the reference's test.py (flags :3-7, helpers :24-121, main :124-218) — BASELINE
config 0's script — on the MI355X path: `model.model(is_training=False)` (ResNet-v1-50 + PixelLink
heads), softmax scores, the script's own `pixel_detect`, one `cv2.minAreaRect` box per
`cv2.findContours` contour, `order_points`, `res_<name>.txt`.

    python test.py --test_data_path ./exhibition --checkpoint_path /tmp/east_icdar2015_resnet_v1_50_rbox/ --output_dir /tmp/res/

Same flag names / defaults.  Everything per-pixel runs on the GPU (network, cv2.resize, softmaxes,
mask, region labelling, convex hulls + rotating calipers); the per-box integer arithmetic of
:191-199 and `order_points` stay on the host as in the reference.  Forced differences: images are
decoded with PIL (or .npy) and no visualisation images are written (`cv2.imwrite` of score_map.jpg /
img.jpg / the annotated photo).  Boxes come out in the order of OpenCV's contour list."""
import argparse
import os
import time

import numpy as np
import torch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--input_size', type=int, default=512)
    ap.add_argument('--gpu_list', type=str, default='1')
    ap.add_argument('--test_data_path', type=str, default='./exhibition')
    ap.add_argument('--checkpoint_path', type=str, default='/tmp/east_icdar2015_resnet_v1_50_rbox/')
    ap.add_argument('--output_dir', type=str, default='/tmp/res/')
    ap.add_argument('--precision', choices=['f16', 'f32'], default='f16', help="f32: the forward pass in the f32 inference "
                    "precision (f32 storage, matrix-core f32 convolutions): score maps within 1e-3 of the f32 reference, ~10x the time")
    return ap.parse_args()


def order_points(pts):
    """test.py:24-35: top-left, top-right, bottom-right, bottom-left."""
    from scipy.spatial import distance as dist
    x_sorted = pts[np.argsort(pts[:, 0]), :]
    left_most = x_sorted[:2, :]
    right_most = x_sorted[2:, :]
    left_most = left_most[np.argsort(left_most[:, 1]), :]
    (tl, bl) = left_most
    D = dist.cdist(tl[np.newaxis], right_most, 'euclidean')[0]
    (br, tr) = right_most[np.argsort(D)[::-1], :]
    return np.array([tl, tr, br, bl], dtype='int32')


def sort_poly(p):
    """test.py:37-43 (unused by main, kept for parity)."""
    min_axis = np.argmin(np.sum(p, axis=1))
    p = p[[min_axis, (min_axis + 1) % 4, (min_axis + 2) % 4, (min_axis + 3) % 4]]
    if abs(p[0, 0] - p[1, 0]) > abs(p[0, 1] - p[1, 1]):
        return p
    return p[[0, 3, 2, 1]]


def pixel_detect(score_map, geo_map, score_map_thresh=0.8, link_thresh=0.8, graph=None):
    """test.py:45-74 (see tool/pixellink_fn.east_pixel_detect)."""
    from tensorflow_ocr_amd.tool import pixellink_fn
    return pixellink_fn.east_pixel_detect(score_map, geo_map, score_map_thresh, link_thresh, graph=graph)


def get_images(test_data_path):
    """test.py:76-90 (+ .npy arrays)."""
    files = []
    exts = ['jpg', 'png', 'jpeg', 'JPG', 'npy']
    for parent, dirnames, filenames in os.walk(test_data_path):
        for filename in sorted(filenames):
            for ext in exts:
                if filename.endswith(ext):
                    files.append(os.path.join(parent, filename))
                    break
    print('Find {} images'.format(len(files)))
    return files


def resize_image(im, max_side_len=3000, graph=None):
    """test.py:92-121: limit the longer side, round both sides to multiples of 32 (the reference's
    rule: already-multiples stay, others become (x // 32 - 1) * 32), cv2.resize.  Returns the resized
    image as a DEVICE float32 [H,W,3] tensor and (ratio_h, ratio_w)."""
    from tensorflow_ocr_amd import ops
    from tensorflow_ocr_amd.graph import get_default_graph
    g = graph or get_default_graph()
    h, w, _ = im.shape
    resize_w, resize_h = w, h
    if max(resize_h, resize_w) > max_side_len:
        ratio = float(max_side_len) / resize_h if resize_h > resize_w else float(max_side_len) / resize_w
    else:
        ratio = 1.
    resize_h = int(resize_h * ratio)
    resize_w = int(resize_w * ratio)
    resize_h = resize_h if resize_h % 32 == 0 else (resize_h // 32 - 1) * 32
    resize_w = resize_w if resize_w % 32 == 0 else (resize_w // 32 - 1) * 32
    if resize_h <= 0 or resize_w <= 0:
        raise ValueError('image too small for the reference sizing rule: %dx%d' % (h, w))
    src = torch.from_numpy(np.ascontiguousarray(im, dtype=np.uint8)).to(g.device)
    out = torch.empty((int(resize_h), int(resize_w), 3), dtype=torch.float32, device=g.device)
    ops.resize_linear_u8(src, out)
    return out, (resize_h / float(h), resize_w / float(w))


def boxes_from_mask(score_map_res, ratio_h, ratio_w, graph=None):
    """test.py:182-199: contours -> minAreaRect -> boxPoints -> np.int0 -> x4 -> / ratio (integer
    array: the division result is truncated on assignment)."""
    from tensorflow_ocr_amd.tool import pixellink_fn
    _, boxes = pixellink_fn.find_contour_boxes(score_map_res, graph=graph)
    out = []
    for box in boxes:
        box = box.copy()
        box[:, 0] = box[:, 0] * 4
        box[:, 1] = box[:, 1] * 4
        box[:, 0] = box[:, 0] / ratio_w
        box[:, 1] = box[:, 1] / ratio_h
        out.append(box)
    return out


def main():
    FLAGS = parse()
    from tensorflow_ocr_amd import checkpoint
    from tensorflow_ocr_amd.datasets import icdar
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.infer import GraphedForward
    from tensorflow_ocr_amd.nets import model
    from tensorflow_ocr_amd.tool import pixellink_fn
    os.makedirs(FLAGS.output_dir, exist_ok=True)
    g = Graph('cuda:0', precision=FLAGS.precision)
    restored = False

    def network(gr, im):      # sess.run([f_score, f_geometry]) + the two softmaxes: one HIP graph per image shape
        f_score, f_geometry = model.model(im, is_training=False, graph=gr)
        fg = f_geometry.data if hasattr(f_geometry, 'data') and not isinstance(f_geometry, torch.Tensor) else f_geometry
        return (pixellink_fn.pixel_scores(f_score, graph=gr),
                pixellink_fn.pixel_scores(fg.reshape(-1, 2), graph=gr).reshape(fg.shape))
    forward = GraphedForward(g, network)
    for im_fn in get_images(FLAGS.test_data_path):
        im = icdar.read_image_rgb(im_fn)                      # cv2.imread(im_fn)[:, :, ::-1]
        start_time = time.time()
        im_resized, (ratio_h, ratio_w) = resize_image(im, graph=g)
        x = im_resized[None]
        scores, pixel_score = forward(x)
        if not restored:
            # variable_averages.variables_to_restore(): the EMA shadows (test.py:149-158)
            if not (os.path.exists(os.path.join(FLAGS.checkpoint_path, 'checkpoint')) or os.path.exists(FLAGS.checkpoint_path + '.index')):
                # the reference dies here too (`ckpt_state.model_checkpoint_path` on None, test.py:155-157):
                # never write boxes produced by randomly initialised weights
                raise FileNotFoundError('no checkpoint under --checkpoint_path %r' % FLAGS.checkpoint_path)
            sd, _ = checkpoint.load_tf_checkpoint(FLAGS.checkpoint_path, use_moving_averages=True)
            g.store.load_state_dict(checkpoint.tf_to_internal(g.store.order, sd), strict=False)
            print('Restore from {}'.format(FLAGS.checkpoint_path))
            scores, pixel_score = forward(x)
            restored = True
        cls_score = scores[:, :, :, 1:2].contiguous()           # softmax(f_score)[..., 1:2]; pixel_score: softmax over the pairs
        torch.cuda.synchronize()
        print('net time:' + str((time.time() - start_time) * 1000) + 'ms')
        score_map_res = pixel_detect(score_map=cls_score, geo_map=pixel_score, graph=g)
        boxes = boxes_from_mask(score_map_res, ratio_h, ratio_w, graph=g)
        res_file = os.path.join(FLAGS.output_dir, 'res_{}.txt'.format(os.path.basename(im_fn).split('.')[0]))
        with open(res_file, 'w') as f:
            for box in boxes:
                box = order_points(box)
                f.write('{},{},{},{},{},{},{},{}\r\n'.format(box[0, 0], box[0, 1], box[1, 0], box[1, 1],
                                                            box[2, 0], box[2, 1], box[3, 0], box[3, 1]))


if __name__ == '__main__':
    main()
