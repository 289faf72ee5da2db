"""Dev-only: reproduce the order-dependent NaN of train_pixellink inside a long-lived process."""
import sys, os, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tensorflow_ocr_amd.graph import Graph
d = tempfile.mkdtemp()
rng = np.random.default_rng(1)
for i in range(6):
    H, W = int(rng.integers(100, 180)), int(rng.integers(100, 180))
    np.save(os.path.join(d, "img_%d.npy" % i), rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8))
    with open(os.path.join(d, "gt_img_%d.txt" % i), "w") as f:
        for k in range(3):
            x0, y0 = int(rng.integers(5, W - 60)), int(rng.integers(5, H - 40))
            w, h = int(rng.integers(20, 50)), int(rng.integers(12, 30))
            f.write("%d,%d,%d,%d,%d,%d,%d,%d,%s\n" % (x0, y0, x0 + w, y0, x0 + w, y0 + h, x0, y0 + h, "###" if k == 2 else "text"))
mode = os.environ.get("PRE", "contours")
if mode == "contours":
    from tensorflow_ocr_amd.tool import pixellink_fn as P
    g0 = Graph("cuda:0")
    m = (np.random.default_rng(0).uniform(size=(96, 128)) < 0.3).astype(np.uint8)
    P.find_contour_boxes(m, graph=g0)
elif mode == "graph":
    g0 = Graph("cuda:0")
elif mode == "cuda":
    torch.zeros(4, device="cuda")
sys.argv = ["train_pixellink.py", "--dataset_dir", d, "--batch_size", "2", "--num_gpus", "1", "--train_image_width", "128",
            "--train_image_height", "128", "--max_number_of_steps", "8", "--log_every_n_steps", "1",
            "--lr_breakpoints", "3,6,9", "--lr_decays", "0.1,0.01,0.001"]
import train_pixellink
train_pixellink.main()
