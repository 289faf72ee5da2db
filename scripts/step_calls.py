#!/usr/bin/env python3
"""Call-by-call timeline of the recorded headline step: every C-ABI call of the replayed plan, in order, with its
entry point, the kernel variant / phase tag, the conv descriptor where there is one, and its device time (HIP events
around the call, median over the timed replays).  What rocprofv3's per-KERNEL table cannot show: which LAYER a
launch belongs to.

    python3 scripts/step_calls.py [--which vgg|pixellink|resnet --batch 64 --size 640] [--reps 7] [--top 40] > gpurun_out/step_calls.json

Events around every call serialise nothing (one stream) but add ~2 us of host work per call: the sum of the call
times is the kernel time of the step, not its wall time."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def build(which, batch, size, device):
    from tensorflow_ocr_amd import synthetic
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import model_vgg_16 as M
    from tensorflow_ocr_amd.train import AdamOptimizer, TrainStep
    g = Graph(device, loss_scale=1024.0, seed=1)
    rng = np.random.default_rng(100)
    if which == "vgg":
        arrs = synthetic.make_batch(rng, batch, size)
        data = [torch.from_numpy(a).to(device) for a in arrs]

        def fl(gr, im, px, lk, mk):
            f_score, f_geometry = M.model_vgg(im, is_training=True, graph=gr)
            return M.loss(px, f_score, lk, f_geometry, mk, graph=gr)
        step = TrainStep(g, fl, lambda gr: AdamOptimizer(gr, learning_rate=1e-4), world_size=1)
    elif which == "resnet":                         # BASELINE configs[3]'s per-GPU share (run with OCR_STORAGE=bf16)
        data = [torch.from_numpy(a).to(device) for a in synthetic.make_batch(rng, batch, size)]

        def fr(gr, im, sm, gm, tm):
            a, b = M.model(im, is_training=True, graph=gr)
            return M.loss(sm, a, gm, b, tm, graph=gr)
        step = TrainStep(g, fr, lambda gr: AdamOptimizer(gr, learning_rate=1e-4), world_size=1)
    elif which == "pixellink":                      # BASELINE configs[2]
        from tensorflow_ocr_amd.nets import pixellink
        from tensorflow_ocr_amd.train import MomentumOptimizer
        data = [torch.from_numpy(a).to(device) for a in synthetic.make_batch(rng, batch, size)]
        data = [(data[0] - 120.0) / 60.0] + data[1:]

        def fp(gr, im, sm, gm, tm):
            return pixellink.PixelLinkNet(im, graph=gr).build_loss(sm[..., 0], gm)
        step = TrainStep(g, fp, lambda gr: MomentumOptimizer(gr), world_size=1)
    else:
        raise SystemExit("unknown --which " + which)
    return step, data


def desc_of(args):
    from tensorflow_ocr_amd import _lib as L
    for a in args[:2]:
        obj = getattr(a, "_obj", None)
        if isinstance(obj, L.ConvDesc):
            return {f: int(getattr(obj, f)) for f, _ in obj._fields_}
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="vgg")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--top", type=int, default=40)
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    step, data = build(args.which, args.batch, args.size, device)
    for _ in range(5):
        step(*data)
    torch.cuda.synchronize()
    plan = step.plan
    assert plan is not None, "the step did not record a plan"
    from tensorflow_ocr_amd import _lib
    times = [[] for _ in plan]
    for _ in range(args.reps):
        evs = []
        for dst, src in zip(step.static_batch, data):
            if src is not dst:
                dst.copy_(src, non_blocking=True)
        for i, e in enumerate(plan):
            if e[0] in ("fork", "join"):          # train.schedule_guests' markers: this walk is serial, one stream
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if e[0] == "c":
                tag = e[4]
                if tag is not None and tag[0] == "xchg":
                    continue
                rc = e[1](*e[2])
                if rc != 0:
                    _lib.check(rc, e[3])
            else:
                e[1]()
            e1.record()
            evs.append((i, e0, e1))
        torch.cuda.synchronize()
        for i, e0, e1 in evs:
            times[i].append(e0.elapsed_time(e1) * 1e3)
    rows = []
    for i, e in enumerate(plan):
        if not times[i]:
            continue
        us = float(np.median(times[i]))
        row = {"i": i, "us": round(us, 1), "call": e[3] if e[0] == "c" else "py:" + (e[2] if len(e) > 2 and isinstance(e[2], str) else "hook")}
        if e[0] == "c":
            tag = e[4]
            if tag is not None:
                row["variant"], row["phase"] = tag[0], (tag[2] if len(tag) > 2 else "")
                if len(tag) > 1 and isinstance(tag[1], float) and tag[1] > 0:
                    row["tflops"] = round(tag[1] / us / 1e6, 1)
            d = desc_of(e[2])
            if d is not None:
                row["conv"] = "%dx%d %d->%d k%d s%d f%d" % (d["h"], d["w"], d["cin"], d["cout"], d["kh"], d["stride"], d["flags"])
        rows.append(row)
    total = sum(r["us"] for r in rows)
    by_call = {}
    for r in rows:
        k = r.get("variant") or r["call"]
        a = by_call.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += r["us"]
    out = {"which": args.which, "calls": len(rows), "sum_us": round(total, 1),
           "by_kernel": [{"kernel": k, "calls": v[0], "us": round(v[1], 1), "share": round(v[1] / total, 4)}
                         for k, v in sorted(by_call.items(), key=lambda kv: -kv[1][1])],
           "timeline": rows}
    print(json.dumps(out, indent=1))
    top = sorted(rows, key=lambda r: -r["us"])[:args.top]
    for r in top:
        print("%4d %8.1f us  %-34s %-30s %-8s %-24s %s" % (r["i"], r["us"], r["call"], r.get("variant", ""), r.get("phase", ""),
                                                          r.get("conv", ""), r.get("tflops", "")), file=sys.stderr)


if __name__ == "__main__":
    main()
