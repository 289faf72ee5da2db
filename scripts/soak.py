"""Soak run: the three training drivers for a few hundred steps each on a small synthetic ICDAR directory
(feeder thread + recorded step + lr schedule + checkpoints), every logged loss finite.

    python scripts/soak.py [steps]
"""
import os, sys, tempfile, io, contextlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
d = tempfile.mkdtemp()
rng = np.random.default_rng(7)
for i in range(24):
    H, W = int(rng.integers(120, 260)), int(rng.integers(120, 260))
    np.save(os.path.join(d, "img_%d.npy" % i), rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8))
    with open(os.path.join(d, "gt_img_%d.txt" % i), "w") as f:
        for k in range(int(rng.integers(1, 6))):
            x0, y0 = int(rng.integers(5, W - 70)), int(rng.integers(5, H - 45))
            w, h = int(rng.integers(20, 60)), int(rng.integers(12, 35))
            f.write("%d,%d,%d,%d,%d,%d,%d,%d,%s\n" % (x0, y0, x0 + w, y0, x0 + w, y0 + h, x0, y0 + h, "###" if k == 4 else "text"))


def run(module, argv):
    import importlib
    mod = importlib.import_module(module)
    old = sys.argv
    sys.argv = [module + ".py"] + argv
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            mod.main()
    finally:
        sys.argv = old
    return buf.getvalue()


out = run("train_pixellink", ["--dataset_dir", d, "--batch_size", "4", "--num_gpus", "1", "--train_image_width", "192",
                              "--train_image_height", "192", "--max_number_of_steps", str(steps), "--log_every_n_steps", "25",
                              "--lr_breakpoints", "100,200,400", "--lr_decays", "1.0,0.5,0.1", "--learning_rate", "0.001"])
losses = [float(l.split("loss = ")[1].split(" ")[0]) for l in out.splitlines() if l.startswith("global step")]
print("train_pixellink", len(losses), "logs, first/last", losses[0], losses[-1], flush=True)
assert len(losses) == (steps + 24) // 25 and all(np.isfinite(losses)), losses
for net in ("model_vgg", "east", "pixellink", "model"):
    out = run("multigpu_train", ["--gpu_list", "0", "--batch_size_per_gpu", "4", "--input_size", "192", "--max_steps", str(steps),
                                 "--net", net, "--num_readers", "2", "--training_data_path", d,
                                 "--checkpoint_path", os.path.join(d, "ckpt_" + net), "--save_checkpoint_steps", "100"])
    losses = [float(l.split("model loss ")[1].split(",")[0]) for l in out.splitlines() if l.startswith("Step ")]
    print("multigpu_train", net, len(losses), "logs, first/last", losses[0], losses[-1], flush=True)
    assert len(losses) == (steps + 9) // 10 and all(np.isfinite(losses)), losses
print("soak ok")
