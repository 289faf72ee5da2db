"""Dev-only: LANMS on bench-like input (MODE=bench) or on pairwise-disjoint quads (MODE=disjoint)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorflow_ocr_amd.graph import Graph
from tensorflow_ocr_amd.tool import lanms
g = Graph("cuda:0")
n, K, S = 16, 1024, 1024
rng = np.random.default_rng(4)
boxes = np.zeros((n, K, 9), np.float32)
mode = os.environ.get("MODE", "bench")
for i in range(n):
    cx = np.sort(rng.uniform(20, S - 20, K)); cy = rng.uniform(20, S - 20, K)
    if mode == "disjoint":
        cy = np.arange(K) * 100.0
    w = rng.uniform(20, 60, K); h = rng.uniform(10, 30, K)
    boxes[i, :, 0] = cx - w; boxes[i, :, 1] = cy - h; boxes[i, :, 2] = cx + w; boxes[i, :, 3] = cy - h
    boxes[i, :, 4] = cx + w; boxes[i, :, 5] = cy + h; boxes[i, :, 6] = cx - w; boxes[i, :, 7] = cy + h
    boxes[i, :, 8] = rng.uniform(0.5, 1.0, K)
if mode == "cluster":          # the detector's regime: every text line fires on dozens of neighbouring pixels
    for i in range(n):
        centers = rng.uniform(40, S - 40, size=(K // 12, 2))
        out = []
        for q in range(K):
            c = centers[rng.integers(len(centers))] + rng.normal(0, 1.5, 2)
            w, h = rng.uniform(40, 90), rng.uniform(12, 24)
            ang = rng.uniform(-0.3, 0.3)
            R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
            pts = (np.array([[-w, -h], [w, -h], [w, h], [-w, h]]) / 2) @ R.T + c
            out.append(np.concatenate([pts.ravel(), [rng.uniform(0.5, 1.0)]]))
        a = np.array(out, np.float32)
        boxes[i] = a[np.lexsort((a[:, 0], a[:, 1].round(-1)))]
bt = torch.from_numpy(boxes).to("cuda:0"); ct = torch.full((n,), K, dtype=torch.int32, device="cuda:0")
for _ in range(6):
    out = lanms.lanms_batch(bt, ct, 0.2, graph=g)
torch.cuda.synchronize()
print(mode, "n_merged", out[1].tolist()[:4], "n_keep", out[3].tolist()[:4])

