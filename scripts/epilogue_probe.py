#!/usr/bin/env python3
"""Is the epilogue of the one-wave-per-SIMD convolution (conv3x3_w4_kernel) bound by the CU or by the chip?  All
workgroups of a round start together and run tiles of equal length, so their epilogues coincide: 256 x 128 KB of
stores (+ as many operand loads in the input-gradient form) hit HBM at once.  This probe runs the same tile on grids of
2048 / 256 / 64 / 4 workgroups (diagnostic build: in-kernel stamps around the main loop and the whole workgroup) and
prints prologue / main loop / epilogue cycles for each: an epilogue that shrinks with the grid is waiting for the chip.

    python3 scripts/epilogue_probe.py > gpurun_out/epilogue_probe.json
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["OCR_HIP_LIB"] = os.environ.get("OCR_DIAG_LIB") or os.path.join(ROOT, "tensorflow_ocr_amd", "libocr_hip_diag.so")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from tensorflow_ocr_amd import _lib as L  # noqa: E402


def read(reader, slots):
    buf = (ctypes.c_ulonglong * (2 * slots))()
    fn = getattr(L.load(), reader)
    fn.restype = ctypes.c_int
    assert fn(buf, ctypes.c_int(slots)) == 0
    return np.frombuffer(buf, dtype=np.uint64).reshape(slots, 2).astype(np.float64)


def run(kind, B, h, w, cin, cout):
    dev = "cuda"
    x = torch.randn(B, h, w, cin, device=dev).half()
    wt = (torch.randn(9, cout, cin, device=dev) * 0.05).half()
    d = L.ConvDesc(B, h, w, cin, h, w, cout, 3, 3, 1, 1, 1, 1, 0, L.CONV_STATS)
    y = torch.empty(B, h, w, cout, dtype=torch.half, device=dev)
    mt = L.call_int("ocr_conv2d_num_mtiles", ctypes.byref(d))
    st = torch.zeros(mt, 2, cout, device=dev)
    if kind == "conv+bnred":
        by = torch.randn(B, h, w, cout, device=dev).half()
        sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
        mu, istd = torch.randn(cout, device=dev) * 0.1, torch.rand(cout, device=dev) + 0.5
        f = lambda: L.call("ocr_conv2d_bnred_f16", ctypes.byref(d), L.ptr(x), L.ptr(wt), L.ptr(y), L.ptr(st), L.ptr(by),
                           L.ptr(sc), L.ptr(sh), L.ptr(mu), L.ptr(istd), ctypes.c_int(1), ctypes.c_int(0), L.stream_ptr())
    else:
        f = lambda: L.call("ocr_conv2d_f16", ctypes.byref(d), L.ptr(x), L.ptr(wt), L.ptr(None), L.ptr(y), L.ptr(st), L.stream_ptr())
    name = ctypes.create_string_buffer(128)
    L.load().ocr_conv2d_variant(ctypes.byref(d), name, ctypes.c_size_t(128))
    wgs = mt * (cout // 256)
    for _ in range(300):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        f()
    e1.record()
    torch.cuda.synchronize()
    slots = min(wgs, 1024)
    a = read("ocr_diag_read_conv", slots)
    b = read("ocr_diag_read_conv_wg", slots)
    main = float(np.median(a[:, 0]))
    ghz = float(np.median(a[:, 0] / a[:, 1] * 0.1))
    wg, pro = float(np.median(b[:, 0])), float(np.median(b[:, 1]))
    return {"kernel": name.value.decode(), "workgroups": wgs, "launch_us": round(e0.elapsed_time(e1) * 20, 1),
            "clock_ghz": round(ghz, 3), "prologue_cycles": round(pro), "main_loop_cycles": round(main),
            "epilogue_cycles": round(wg - main - pro), "workgroup_cycles": round(wg)}


def main():
    out = {}
    for label, (h, w, cin, cout) in {"conv3_2 256->256": (128, 128, 256, 256), "conv4_2 512->512": (64, 64, 512, 512)}.items():
        for kind in ("conv", "conv+bnred"):
            per_img = (h // 8) * (w // 32) * (cout // 256)
            for B, hh, ww in ((32, h, w), (256 // per_img, h, w), (64 // per_img, h, w), (1, 32, 64), (1, 8, 32)):
                r = run(kind, B, hh, ww, cin, cout)
                out["%s | %s | n=%d %dx%d" % (label, kind, B, hh, ww)] = r
                print(label, kind, B, hh, ww, r, file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
