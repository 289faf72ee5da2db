"""GPU: the bfloat16 build (libocr_hip_bf16.so, OCR_STORAGE=bf16 — BASELINE.json configs[3] "EAST
ResNet-v1-50 ... bf16").  The storage type is a process-wide choice, so the bf16 checks run in ONE
child interpreter: the conv forward / input-gradient / weight-gradient sweep over every tile variant,
every single-layer parity test, the ResNet blocks and both ResNet graphs, the VGG model end to end,
PixelLinkNet and the train-step tests (replay == eager, Adam/EMA, two ranks in sync), all against the
oracle rounding to bfloat16 at the same storage points (tolerances in those tests: 8x the f16 bars =
the ratio of the two roundings; end to end: the net's own f32-vs-bf16 sensitivity)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bf16_build_layers_models_and_train_step(device):
    env = dict(os.environ, OCR_STORAGE="bf16")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_conv_abi.py"),
                        os.path.join(ROOT, "tests", "test_gpu_layers.py"),
                        os.path.join(ROOT, "tests", "test_gpu_resnet.py"),
                        os.path.join(ROOT, "tests", "test_gpu_model_vgg.py"),
                        os.path.join(ROOT, "tests", "test_gpu_pixellink.py"),
                        os.path.join(ROOT, "tests", "test_gpu_train_step.py")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    import re
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 45 and "failed" not in r.stdout, tail


def test_bf16_resnet50_east_640_batch64_parity(device):
    """BASELINE configs[3] as quoted (bf16, batch 64, 640^2): n = 64 replicated == n = 2 in the bf16 library."""
    env = dict(os.environ, OCR_STORAGE="bf16")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-s", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_batch_parity.py"), "-k", "bf16_batch64"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    print("\n".join(l for l in r.stdout.splitlines() if l.startswith("bf16 n=64")))
    assert r.returncode == 0 and "1 passed" in r.stdout, tail
