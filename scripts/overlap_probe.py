#!/usr/bin/env python3
"""Does a small persistent kernel on a second stream co-execute with the step's conv kernels, or do the two time-slice?
(The question behind bench.py's `exchange.proxy`.)  A = 24 launches of conv4_2 forward (conv3x3_w4_kernel: one 512-VGPR wave
per SIMD) — and, separately, of an HBM-bound bn_relu pass — on the current stream; B = ocr_comm_proxy (N workgroups, paced)
on another stream for about the same wall time.  Reports A alone, A with B, for plain / high-priority second streams."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))


def main():
    import ctypes
    from tensorflow_ocr_amd import _lib as L, ops
    from tensorflow_ocr_amd.graph import F16
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    n, h, w, c = 32, 64, 64, 512
    x = torch.from_numpy(rng.standard_normal((n, h, w, c)).astype(np.float32)).to(dev).to(F16)
    wt = torch.from_numpy((rng.standard_normal((3, 3, c, c)) / 68).astype(np.float32)).to(dev)
    kc, ck = torch.empty((9, c, c), dtype=F16, device=dev), torch.empty((9, c, c), dtype=F16, device=dev)
    ops.pack_weights(wt, kc, ck)
    y = torch.empty((n, h, w, c), dtype=F16, device=dev)
    d = ops.conv_desc((n, h, w, c), c, 3, 3)
    scale = torch.ones(c, device=dev)
    shift = torch.zeros(c, device=dev)
    big = torch.empty((32, 256, 256, 128), dtype=F16, device=dev).normal_()
    big_o = torch.empty_like(big)
    buf = torch.zeros(64 << 20, dtype=torch.uint8, device=dev)
    stats = torch.tensor([-1, 0, 0, 0, 0, 0, 0, 0], dtype=torch.int64).to(dev)

    def conv_a():
        for _ in range(24):
            ops.conv2d(d, x, kc, y)

    def bn_a():
        for _ in range(24):
            ops.bn_relu(big, scale, shift, True, 0, big_o, None)

    def proxy(stream, wg, gbps, nbytes):
        L.call("ocr_comm_proxy", L.ptr(buf), ctypes.c_size_t(nbytes), ctypes.c_int(wg), ctypes.c_float(gbps), L.ptr(stats),
               ctypes.c_void_p(stream.cuda_stream))

    # the kernels whose grids fill the chip in exactly ONE round of workgroups: the 3x3 weight gradient (split-K sized to
    # the CU count) and the persistent 64-channel kernel (conv1_2)
    from tensorflow_ocr_amd.ops import Workspace
    ws = Workspace(dev, 256 << 20)
    dy = torch.empty((n, h, w, c), dtype=F16, device=dev).normal_()
    dw = torch.empty((3, 3, c, c), dtype=torch.float32, device=dev)

    def wgrad_a():
        for _ in range(24):
            ops.conv2d_wgrad(d, x, dy, dw, ws)
    x1 = torch.empty((8, 512, 512, 64), dtype=F16, device=dev).normal_()
    w1 = torch.from_numpy((rng.standard_normal((3, 3, 64, 64)) / 24).astype(np.float32)).to(dev)
    kc1, ck1 = torch.empty((9, 64, 64), dtype=F16, device=dev), torch.empty((9, 64, 64), dtype=F16, device=dev)
    ops.pack_weights(w1, kc1, ck1)
    y1 = torch.empty_like(x1)
    d1 = ops.conv_desc((8, 512, 512, 64), 64, 3, 3)

    def c64_a():
        for _ in range(24):
            ops.conv2d(d1, x1, kc1, y1)
    out = {"variants": {"conv4_2_fwd": ops.conv2d_variant(d), "conv1_2_fwd": ops.conv2d_variant(d1)}}
    for name, fa in (("conv4_2_fwd_x24", conv_a), ("conv4_2_wgrad_x24", wgrad_a), ("conv1_2_fwd_n8_x24", c64_a),
                     ("bn_relu_512MiB_x24", bn_a)):
        fa()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fa(); e1.record(); torch.cuda.synchronize()
        alone = e0.elapsed_time(e1)
        res = {"alone_ms": round(alone, 3)}
        for sname, st in (("plain_stream", torch.cuda.Stream()), ("high_priority_stream", torch.cuda.Stream(priority=-1))):
            for wg in (24,):
                # B sized to last about as long as A: 2 * bytes / rate = alone
                gbps = 150.0
                nbytes = int(min(buf.numel(), alone * 1e-3 * gbps * 1e9 / 2)) // 16 * 16
                torch.cuda.synchronize()
                b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                st.wait_stream(torch.cuda.current_stream())
                b0.record(st)
                proxy(st, wg, gbps, nbytes)
                b1.record(st)
                e0.record(); fa(); e1.record()
                torch.cuda.synchronize()
                res["%s_wg%d" % (sname, wg)] = {"a_ms": round(e0.elapsed_time(e1), 3), "b_ms": round(b0.elapsed_time(b1), 3),
                                                 "b_paced_ms": round(2 * nbytes / gbps / 1e6, 3)}
        out[name] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
