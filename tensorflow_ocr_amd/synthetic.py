"""Synthetic training batches of the shape the reference's icdar generator feeds
(multigpu_train.py:164-174: images [N,S,S,3] 0..255, score map [N,S/4,S/4,1], 8 link maps,
training mask), per SURVEY.md §8d: pixel label = union of random axis-aligned rectangles at 1/4
resolution, link label = 1 where the neighbour in that direction shares the rectangle (border = 1;
direction order left, left_down, left_up, right, right_down, right_up, up, down), mask = 1."""
import numpy as np


def make_batch(rng, n, size, rects=8):
    q4 = size // 4
    images = rng.uniform(0, 255, size=(n, size, size, 3)).astype(np.float32)
    ids = np.zeros((n, q4, q4), np.int32)
    for b in range(n):
        for k in range(rects):
            hh = int(rng.integers(max(2, q4 // 16), max(3, q4 * 3 // 8)))
            ww = int(rng.integers(max(2, q4 // 16), max(3, q4 * 3 // 8)))
            y0 = int(rng.integers(0, q4 - hh + 1))
            x0 = int(rng.integers(0, q4 - ww + 1))
            ids[b, y0:y0 + hh, x0:x0 + ww] = k + 1
    pixel = (ids > 0).astype(np.float32)[..., None]
    offs = [(-1, 0), (-1, 1), (-1, -1), (1, 0), (1, 1), (1, -1), (0, -1), (0, 1)]   # (dx, dy)
    link = np.zeros((n, q4, q4, 8), np.float32)
    pad = np.pad(ids, ((0, 0), (1, 1), (1, 1)), constant_values=-1)
    for d, (dx, dy) in enumerate(offs):
        nb = pad[:, 1 + dy:1 + dy + q4, 1 + dx:1 + dx + q4]
        link[..., d] = ((ids > 0) & ((nb == ids) | (nb == -1))).astype(np.float32)
    mask = np.ones((n, q4, q4, 1), np.float32)
    return images, pixel, link, mask
