#!/usr/bin/env python3
"""In-kernel clock of the two dominant MFMA kernels, on random and on all-zero operands
(MI355X_MICROARCH.md "DVFS give-back" item 6).  OCR_CONV_W4=0 / OCR_CONV_W4S=0 /
OCR_WGRAD3=0 select the round-1 kernels (conv_igemm_kernel<...>, wgrad2_kernel<...>) for comparison.

Uses the DIAGNOSTIC library libocr_hip_diag.so (same sources, -DOCR_DIAG_CLOCK: s_memtime /
s_memrealtime stamped once around the main loop of conv_igemm_kernel and wgrad2_kernel, written to a
buffer of their own).  Every case runs back to back for >= 2 s before the stamps are read; the clock
is the median over workgroups of d(s_memtime) / d(s_memrealtime) x 100 MHz.  Wall TFLOP/s come from HIP
events around the last second of launches.  Never quote this build's run time as the product's.

    python3 scripts/clock_diag.py > profiles/r02_clock_diag.json
"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (OCR_DIAG_LIB: another diagnostic build, e.g. one compiled with a -DW4_ABL ablation; OCR_DIAG_CASES: substring filter)
os.environ["OCR_HIP_LIB"] = os.environ.get("OCR_DIAG_LIB") or os.path.join(ROOT, "tensorflow_ocr_amd", "libocr_hip_diag.so")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from tensorflow_ocr_amd import _lib as L  # noqa: E402

B = 32
# (hw, cin, cout): conv4_2-like (conv_igemm<256,64,4,1,8>, wgrad2<128,9>), conv2_2-like (<128,64,2,1,16>)
CASES = {"conv4_2 64x64 512->512": (64, 512, 512), "conv3_2 128x128 256->256": (128, 256, 256),
         "conv2_2 256x256 128->128": (256, 128, 128)}


def clock(reader, slots):
    buf = (ctypes.c_ulonglong * (2 * slots))()
    fn = getattr(L.load(), reader)
    fn.restype = ctypes.c_int
    assert fn(buf, ctypes.c_int(slots)) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(slots, 2).astype(np.float64)
    ok = a[:, 1] > 0
    ghz = a[ok, 0] / a[ok, 1] * 0.1
    return float(np.median(ghz)), float(np.percentile(ghz, 10)), float(np.percentile(ghz, 90)), float(np.median(a[ok, 0]))


def run_case(kind, hw, cin, cout, zero):
    dev = "cuda"
    x = torch.randn(B, hw, hw, cin, device=dev).half()
    w = (torch.randn(9, cout, cin, device=dev) * 0.05).half()
    dy = torch.randn(B, hw, hw, cout, device=dev).half()
    if zero:
        x.zero_(); w.zero_(); dy.zero_()
    d = L.ConvDesc(B, hw, hw, cin, hw, hw, cout, 3, 3, 1, 1, 1, 1, 0, L.CONV_STATS if kind.startswith("conv") else 0)
    flops = 2.0 * B * hw * hw * cout * cin * 9
    if kind == "conv+bnred":
        # the input-gradient form with the fused BN-backward reduction of the layer below (epilogue mode 2): the
        # epilogue also reads the operand tile `bn_y` and folds (sum dz, sum dz*xhat) per channel
        y = torch.empty(B, hw, hw, cout, dtype=torch.half, device=dev)
        mt = L.call_int("ocr_conv2d_num_mtiles", ctypes.byref(d))
        st = torch.zeros(mt, 2, cout, device=dev)
        by = torch.randn(B, hw, hw, cout, device=dev).half()
        sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
        mu, istd = torch.randn(cout, device=dev) * 0.1, torch.rand(cout, device=dev) + 0.5
        if zero:
            by.zero_()
        f = lambda: L.call("ocr_conv2d_bnred_f16", ctypes.byref(d), L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(st), L.ptr(by),
                           L.ptr(sc), L.ptr(sh), L.ptr(mu), L.ptr(istd), ctypes.c_int(1), ctypes.c_int(0), L.stream_ptr())
        name = ctypes.create_string_buffer(128)
        L.load().ocr_conv2d_variant(ctypes.byref(d), name, ctypes.c_size_t(128))
        variant, reader, slots = name.value.decode(), "ocr_diag_read_conv", 1024
    elif kind == "conv":
        y = torch.empty(B, hw, hw, cout, dtype=torch.half, device=dev)
        mt = L.call_int("ocr_conv2d_num_mtiles", ctypes.byref(d))
        st = torch.zeros(mt, 2, cout, device=dev)
        f = lambda: L.call("ocr_conv2d_f16", ctypes.byref(d), L.ptr(x), L.ptr(w), L.ptr(None), L.ptr(y), L.ptr(st), L.stream_ptr())
        name = ctypes.create_string_buffer(128)
        L.load().ocr_conv2d_variant(ctypes.byref(d), name, ctypes.c_size_t(128))
        variant, reader, slots = name.value.decode(), "ocr_diag_read_conv", 1024
    else:
        nbytes = L.call_size("ocr_conv2d_wgrad_workspace", ctypes.byref(d))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        dw = torch.empty(3, 3, cin, cout, device=dev)
        f = lambda: L.call("ocr_conv2d_wgrad_f16", ctypes.byref(d), L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(ws), ctypes.c_size_t(nbytes), L.stream_ptr())
        v3 = os.environ.get("OCR_WGRAD3", "1") != "0" and cout % 128 == 0 and cin % 64 == 0
        variant = "wgrad3_kernel<9>" if v3 else "wgrad2_kernel<%d,9>" % (128 if cout % 128 == 0 else 64)
        reader, slots = "ocr_diag_read_wgrad", 256
    f()
    torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    while time.time() - t0 < 2.0:                      # >= 2 s of back-to-back launches
        for _ in range(50):
            f()
        torch.cuda.synchronize()
        n += 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    med, p10, p90, cyc = clock(reader, slots)
    wg = None
    if kind.startswith("conv") and variant.startswith(("conv3x3_w4_kernel", "conv3x3_w4s_kernel")):
        # whole-workgroup cycles (kernel entry -> end of wave 0's epilogue instruction stream): tile = main loop + the rest
        buf = (ctypes.c_ulonglong * (2 * slots))()
        fn = getattr(L.load(), reader + "_wg")
        fn.restype = ctypes.c_int
        assert fn(buf, ctypes.c_int(slots)) == 0
        a = np.frombuffer(buf, dtype=np.uint64).reshape(slots, 2).astype(np.float64)
        okw = a[:, 0] > 0
        wg = float(np.median(a[okw, 0]))
        pro = float(np.median(a[okw, 1]))
    extra = {} if wg is None else {"workgroup_cycles_median": round(wg), "prologue_cycles": round(pro),
                                   "epilogue_cycles": round(wg - cyc - pro), "prologue_epilogue_cycles": round(wg - cyc),
                                   "prologue_epilogue_share": round((wg - cyc) / wg, 3),
                                   "prologue_epilogue_us": round((wg - cyc) / (med * 1e3), 2)}
    return {**extra, "kernel": variant, "operands": "zero" if zero else "random", "launch_ms_diag_build": round(ms, 4),
            "tflops_diag_build": round(flops / ms / 1e9, 1), "clock_ghz_median": round(med, 3), "clock_ghz_p10": round(p10, 3),
            "clock_ghz_p90": round(p90, 3), "main_loop_cycles_median": round(cyc)}


def main():
    out = {}
    only = os.environ.get("OCR_DIAG_CASES", "")
    for label, (hw, cin, cout) in CASES.items():
        if only and only not in label:
            continue
        for kind in (("conv",) if only else ("conv", "conv+bnred", "wgrad")):
            if kind == "conv+bnred" and cout % 128:
                continue                                     # the stamped epilogue split exists in the w4 / w4s kernels only
            for zero in (False, True):
                r = run_case(kind, hw, cin, cout, zero)
                out["%s | %s | %s" % (label, kind, r["operands"])] = r
                print(label, kind, r, file=sys.stderr, flush=True)
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from pmc_mfma import provenance
    res = {"_provenance": provenance("python3 scripts/clock_diag.py (libocr_hip_diag.so)")}
    res.update(out)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
