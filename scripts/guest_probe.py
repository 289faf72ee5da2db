#!/usr/bin/env python3
"""What do the library's MFMA kernels and a small-footprint guest on a second stream cost each other?  
Hosts: the real kernels (ops.*) x24 on the current stream.  Guests (scripts/guest_kernels.hip -> scripts/_bin/libguest.so):
spin = 24 or 256 workgroups that hold a slot for 1 ms, no memory traffic; copy = an HBM stream (1 GiB read + 1 GiB written
per launch, 27 VGPRs, 256 workgroups), launched back to back for about the host's duration."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from tensorflow_ocr_amd import _lib as L, ops
    from tensorflow_ocr_amd.graph import F16
    from tensorflow_ocr_amd.ops import Workspace
    G = ctypes.CDLL(os.path.join(ROOT, "scripts", "_bin", "libguest.so"))
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    ws = Workspace(dev, 512 << 20)

    def mk(n, h, w, cin, cout, k=3, dil=1):
        x = torch.empty((n, h, w, cin), dtype=F16, device=dev).normal_()
        wt = (torch.randn((k, k, cin, cout), device=dev) / (k * (cin ** 0.5))).float()
        kc, ck = torch.empty((k * k, cout, cin), dtype=F16, device=dev), torch.empty((k * k, cin, cout), dtype=F16, device=dev)
        ops.pack_weights(wt, kc, ck)
        d = ops.conv_desc((n, h, w, cin), cout, k, k, dilation=dil) if dil != 1 else ops.conv_desc((n, h, w, cin), cout, k, k)
        y = torch.empty((n, d.oh, d.ow, cout), dtype=F16, device=dev)
        dy = torch.empty((n, d.oh, d.ow, cout), dtype=F16, device=dev).normal_()
        dw = torch.empty((k, k, cin, cout), dtype=torch.float32, device=dev)
        return dict(d=d, x=x, kc=kc, y=y, dy=dy, dw=dw)
    c42 = mk(32, 64, 64, 512, 512)
    c32 = mk(32, 128, 128, 256, 256)
    c22 = mk(32, 256, 256, 128, 128)
    c12 = mk(8, 512, 512, 64, 64)
    hosts = {
        "conv4_2_wgrad": lambda: ops.conv2d_wgrad(c42["d"], c42["x"], c42["dy"], c42["dw"], ws),
        "conv3_2_wgrad": lambda: ops.conv2d_wgrad(c32["d"], c32["x"], c32["dy"], c32["dw"], ws),
        "conv2_2_wgrad": lambda: ops.conv2d_wgrad(c22["d"], c22["x"], c22["dy"], c22["dw"], ws),
        "conv4_2_fwd": lambda: ops.conv2d(c42["d"], c42["x"], c42["kc"], c42["y"]),
        "conv2_2_fwd": lambda: ops.conv2d(c22["d"], c22["x"], c22["kc"], c22["y"]),
        "conv1_2_fwd_n8": lambda: ops.conv2d(c12["d"], c12["x"], c12["kc"], c12["y"]),
    }
    variants = {k: ops.conv2d_variant(v["d"]) for k, v in (("conv4_2", c42), ("conv3_2", c32), ("conv2_2", c22), ("conv1_2", c12))}
    src = torch.zeros(1 << 30, dtype=torch.uint8, device=dev)
    dst = torch.zeros(1 << 30, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream()
    NREP = 24

    def run_host(f):
        for _ in range(NREP):
            f()

    def guest(kind, n_launch):
        s = ctypes.c_void_p(side.cuda_stream)
        for _ in range(n_launch):
            if kind == "spin24":
                G.guest_spin(ctypes.c_ulonglong(100000), 24, None, s)
            elif kind == "spin256":
                G.guest_spin(ctypes.c_ulonglong(100000), 256, None, s)
            elif kind == "copy_u4":
                G.guest_copy(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), ctypes.c_size_t(1 << 30), 256, 4, 256, None, s)
            elif kind == "copy_u2x2":
                G.guest_copy(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), ctypes.c_size_t(1 << 30), 512, 2, 256, None, s)

    def timed(fa, gk, gn, guest_first=True):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if gk and guest_first:
            g0.record(side); guest(gk, gn); g1.record(side)
        e0.record()
        if fa:
            run_host(fa)
        e1.record()
        if gk and not guest_first:
            g0.record(side); guest(gk, gn); g1.record(side)
        torch.cuda.synchronize()
        return (e0.elapsed_time(e1) if fa else None), (g0.elapsed_time(g1) if gk else None)

    out = {"variants": variants, "nrep": NREP, "cases": []}
    # guests alone
    galone = {}
    for gk in ("spin24", "copy_u4", "copy_u2x2"):
        timed(None, gk, 2)
        galone[gk] = timed(None, gk, 4)[1] / 4
    out["guest_alone_ms_per_launch"] = galone
    for name, fa in hosts.items():
        run_host(fa)
        alone = timed(fa, None, 0)[0]
        alone = min(alone, timed(fa, None, 0)[0])
        for gk in ("spin24", "spin256", "copy_u4", "copy_u2x2"):
            per = galone.get(gk, 1.0)
            gn = max(1, int(alone * 0.6 / per)) if gk.startswith("copy") else 1
            a, g = timed(fa, gk, gn)
            out["cases"].append({"host": name, "guest": gk, "guest_launches": gn, "host_alone_ms": round(alone, 3), "host_with_ms": round(a, 3),
                                 "guest_alone_ms": round(per * gn, 3), "guest_with_ms": round(g, 3)})
            print(json.dumps(out["cases"][-1]), flush=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "guest_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
