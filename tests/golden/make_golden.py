"""Generates the golden fixtures in this directory FROM THE ORACLE (oracle/ocr_oracle.py and the
OpenCV / pipeline restatements beside it).

The reference itself cannot run (Python 2 / TF 1.4 / cv2 absent) and ships no
vectors, so these pin the oracle against accidental change and give the GPU tests fixed
input/output pairs; they are NOT reference outputs (parity unpinned).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ocr_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def model_vgg_small():
    """Width/8 model_vgg on one 64x64 image, f32: inputs, parameters (seeded), outputs, loss, a few
    gradient norms."""
    rng = np.random.default_rng(7)
    p = O.init_model_vgg_params(rng, width_div=8)
    images, pixel, link, mask = O.synthetic_batch(rng, 1, 64)
    tp = O.to_torch_params(p)
    px, lk, _ = O.model_vgg(torch.from_numpy(images), tp, True, mixed=False)
    L = O.dice_loss(torch.from_numpy(pixel), px, torch.from_numpy(link), lk, torch.from_numpy(mask))
    L.backward()
    out = {"images": images, "pixel": pixel, "link": link, "mask": mask,
           "pixel_cls": px.detach().numpy(), "link_cls": lk.detach().numpy(),
           "loss": np.float32(L.item())}
    for k in ("conv1/conv1_1/weights", "conv5/conv5_3/weights", "fc7/BatchNorm/gamma",
              "feature_fusion/Conv_9/weights"):
        out["gradnorm:" + k] = np.float32(tp[k].grad.norm().item())
    np.savez_compressed(os.path.join(HERE, "model_vgg_w8_64.npz"), **out)


def primitives():
    rng = np.random.default_rng(11)
    x = rng.standard_normal((1, 5, 6, 3)).astype(np.float32)
    up = O.resize_bilinear_x2(torch.from_numpy(x)).numpy()
    xp = rng.standard_normal((2, 7, 9, 4)).astype(np.float32)
    p22 = O.max_pool(torch.from_numpy(xp), 2, 2).numpy()
    p31 = O.max_pool(torch.from_numpy(xp), 3, 1).numpy()
    p32 = O.max_pool(torch.from_numpy(xp), 3, 2).numpy()
    w = rng.standard_normal((3, 3, 4, 5)).astype(np.float32)
    c1 = O.conv2d(torch.from_numpy(xp), torch.from_numpy(w), 1, 1).numpy()
    c6 = O.conv2d(torch.from_numpy(xp), torch.from_numpy(w), 1, 6).numpy()
    cs2 = O.conv2d_same(torch.from_numpy(xp), torch.from_numpy(w), 2).numpy()
    yt = (rng.uniform(size=(2, 8, 8, 1)) < 0.3).astype(np.float32)
    yl = (rng.uniform(size=(2, 8, 8, 8)) < 0.3).astype(np.float32)
    pp = rng.uniform(size=(2, 8, 8, 2)).astype(np.float32)
    pl = rng.uniform(size=(2, 8, 8, 16)).astype(np.float32)
    m = (rng.uniform(size=(2, 8, 8, 1)) < 0.9).astype(np.float32)
    dl = O.dice_loss(*(torch.from_numpy(a) for a in (yt, pp, yl, pl, m))).item()
    np.savez_compressed(os.path.join(HERE, "primitives.npz"), x=x, up=up, xp=xp, p22=p22, p31=p31,
                        p32=p32, w=w, c1=c1, c6=c6, cs2=cs2, yt=yt, yl=yl, pp=pp, pl=pl, m=m,
                        dice=np.float32(dl))


def cv_geometry():
    """Pins of the OpenCV / pipeline restatements (oracle/cvgeom_oracle.c, labels.py, evalboxes.py,
    contours.py): fixed inputs and the outputs they produce today."""
    from oracle import contours as OC
    from oracle import cvgeom as C
    from oracle import evalboxes as OE
    from oracle import labels as OL
    rng = np.random.default_rng(21)
    out = {}
    pts = rng.integers(0, 200, size=(40, 2)).astype(np.int32)
    rect, cal, hull = C.min_area_rect(pts)
    out.update(mar_pts=pts, mar_rect=rect, mar_cal=cal, mar_hull=hull, mar_box=C.box_points(rect))
    quad = np.array([[5, 3], [58, 9], [49, 40], [2, 30]], np.int32)
    out.update(fill_quad=quad, fill_img=C.fill_poly(np.zeros((48, 64), np.uint8), quad, 1))
    src = rng.integers(0, 256, size=(37, 53, 3)).astype(np.uint8)
    out.update(rs_src=src, rs_64=C.resize_linear_u8(src, 64, 64), rs_half=C.resize_linear_u8(src[:36, :52], 18, 26))
    polys = np.array([[[10, 8], [50, 12], [48, 30], [8, 26]], [[30, 20], [60, 22], [58, 44], [28, 40]],
                      [[2, 50], [12, 50], [12, 56], [2, 56]]], np.float32)
    tags = np.array([False, True, False])
    s4, g4, m4 = OL.icdar_labels((64, 64), polys, tags)
    out.update(lab_polys=polys, lab_tags=tags, lab_score=s4, lab_geo=g4, lab_mask=m4)
    ps, pl, _ = OL.pixellink_generate_rbox(64, 64, polys[:, :, 0] / 64, polys[:, :, 1] / 64,
                                           np.zeros((3, 4), np.float32), np.zeros(3, np.int32))
    out.update(pl_score=ps, pl_link=pl)
    det = np.array([[10, 8, 50, 12, 48, 30, 8, 26], [31, 21, 59, 23, 57, 43, 29, 39], [70, 70, 90, 70, 90, 80, 70, 80]])
    n, tp, fp = OE.bboxes_matching(det, polys[:, :, 0].astype(int), polys[:, :, 1].astype(int), np.array([0, 0, 1]))
    out.update(ev_det=det, ev_n=np.int32(n), ev_tp=tp, ev_fp=fp,
               ev_iou=np.stack([OE.np_bboxes_jaccard(d, polys[:, :, 0].astype(int), polys[:, :, 1].astype(int)) for d in det]))
    mask = np.zeros((24, 32), np.uint8)
    mask[3:15, 4:20] = 1
    mask[6:10, 8:12] = 0
    mask[18:22, 25:31] = 1
    rects, boxes = OC.contour_boxes(mask)
    out.update(ct_mask=mask, ct_rects=np.stack(rects), ct_boxes=np.stack(boxes))
    traced = OC.suzuki_contours(mask)
    out.update(ct_kinds=np.array([k for k, _, _ in traced]), ct_parents=np.array([p for _, _, p in traced], np.int32))
    cub = np.random.default_rng(22).uniform(size=(12, 20)).astype(np.float32)
    out.update(cub_src=cub, cub_up=C.resize_cubic_f32(cub * np.float32(255), 45, 80), cub_down=C.resize_cubic_f32(cub, 5, 9))
    np.savez_compressed(os.path.join(HERE, "cv_geometry.npz"), **out)


if __name__ == "__main__":
    model_vgg_small()
    primitives()
    cv_geometry()
    print("golden fixtures written to", HERE)
