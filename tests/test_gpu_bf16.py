"""GPU: the bfloat16 build (libocr_hip_bf16.so, OCR_STORAGE=bf16 — BASELINE.json configs[3] "EAST
ResNet-v1-50 ... bf16").  The storage type is a process-wide choice, so the bf16 checks run in ONE
child interpreter: the conv forward / input-gradient / weight-gradient sweep over every tile variant
and the ResNet + EAST-merge train step, both against the oracle rounding to bfloat16 at the same
storage points (tolerances in those tests: 8x the f16 bars = the ratio of the two roundings)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bf16_build_conv_sweep_and_east_step(device):
    env = dict(os.environ, OCR_STORAGE="bf16")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_conv_abi.py"),
                        os.path.join(ROOT, "tests", "test_gpu_resnet.py") + "::test_model_east_merge_branch_dice",
                        os.path.join(ROOT, "tests", "test_gpu_layers.py") + "::test_storage_dtype_matches_library"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    import re
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 19 and "failed" not in r.stdout, tail
