"""GPU: the reference-script counterparts run end to end on a small ICDAR-style directory through
the device feeder (decode workers -> pinned slab -> resize + label kernels -> recorded train step):
multigpu_train.py (icdar.generate_rbox labels, dice loss) and train_pixellink.py
(pixellink_fn.generate_rbox labels, PixelLink loss, Momentum + staircase LR)."""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dataset(tmp_path, n=6, seed=0):
    rng = np.random.default_rng(seed)
    for i in range(n):
        H, W = int(rng.integers(100, 180)), int(rng.integers(100, 180))
        np.save(os.path.join(tmp_path, "img_%d.npy" % i), rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8))
        with open(os.path.join(tmp_path, "gt_img_%d.txt" % i), "w") as f:
            for k in range(3):
                x0, y0 = int(rng.integers(5, W - 60)), int(rng.integers(5, H - 40))
                w, h = int(rng.integers(20, 50)), int(rng.integers(12, 30))
                f.write("%d,%d,%d,%d,%d,%d,%d,%d,%s\n" % (x0, y0, x0 + w, y0, x0 + w, y0 + h, x0, y0 + h,
                                                        "###" if k == 2 else "text"))
    return str(tmp_path)


def _run(module, argv, capsys):
    sys.path.insert(0, ROOT)
    mod = importlib.import_module(module)
    old = sys.argv
    sys.argv = [module + ".py"] + argv
    try:
        mod.main()
    finally:
        sys.argv = old
    return capsys.readouterr().out


def test_multigpu_train_on_icdar_directory(device, tmp_path, capsys):
    d = _dataset(tmp_path)
    out = _run("multigpu_train", ["--gpu_list", "0", "--batch_size_per_gpu", "2", "--input_size", "128",
                                  "--max_steps", "11", "--net", "model_vgg", "--num_readers", "2",
                                  "--training_data_path", d, "--checkpoint_path", os.path.join(d, "ckpt"),
                                  "--save_checkpoint_steps", "5"], capsys)
    lines = [l for l in out.splitlines() if l.startswith("Step ")]
    assert len(lines) == 2 and lines[0].startswith("Step 000000, model loss ") and "examples/second" in lines[1]
    losses = [float(l.split("model loss ")[1].split(",")[0]) for l in lines]
    assert all(np.isfinite(losses))
    # saver.save every --save_checkpoint_steps: TF V2 bundles with the reference's names + EMA shadows
    from tensorflow_ocr_amd import checkpoint, tf_bundle
    ck = os.path.join(d, "ckpt")
    assert tf_bundle.get_checkpoint_state(ck).endswith("model.ckpt-10")
    sd, step = checkpoint.load_tf_checkpoint(ck)
    assert step == 10 and sd["conv1/conv1_1/weights"].shape == (3, 3, 3, 64)
    assert "conv5/conv5_3/BatchNorm/moving_variance" in sd
    keys = [k.decode() for k, _ in tf_bundle.read_table(tf_bundle.get_checkpoint_state(ck) + ".index")]
    assert "conv1/conv1_1/weights/ExponentialMovingAverage" in keys and "global_step" in keys
    out = _run("multigpu_train", ["--gpu_list", "0", "--batch_size_per_gpu", "2", "--input_size", "128",
                                  "--max_steps", "1", "--net", "model_vgg", "--num_readers", "0", "--restore",
                                  "--training_data_path", os.path.join(d, "none"), "--checkpoint_path", ck], capsys)
    assert "continue training from previous checkpoint" in out


def test_train_pixellink_on_icdar_directory(device, tmp_path, capsys):
    d = _dataset(tmp_path, seed=1)
    out = _run("train_pixellink", ["--dataset_dir", d, "--batch_size", "2", "--num_gpus", "1",
                                   "--train_image_width", "128", "--train_image_height", "128",
                                   "--max_number_of_steps", "8", "--log_every_n_steps", "4",
                                   "--lr_breakpoints", "3,6,9", "--lr_decays", "0.1,0.01,0.001"], capsys)
    lines = [l for l in out.splitlines() if l.startswith("global step")]
    assert len(lines) == 2
    assert "lr 0.001000" in lines[0] and "lr 0.000100" in lines[1]          # 0.01 * 0.1, then * 0.01
    assert all(np.isfinite(float(l.split("loss = ")[1].split(" ")[0])) for l in lines)
