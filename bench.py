#!/usr/bin/env python3
"""Headline benchmark: images/sec of one data-parallel TRAINING step (forward + dice loss +
backward + gradient all-reduce + Adam/EMA) of the VGG-16 detector at 512x512, batch 32 per GPU
(BASELINE.json configs[1]; the graph is nets/model_vgg_16.py: model_vgg + loss).

    python bench.py --gpus N --steps K --warmup W        (N > 1: spawns the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel:
the implicit-GEMM conv, achieved TFLOP/s from HIP-event timing of every launch inside the timed
steps) and, at N=1, `cpu_baseline` (the CPU oracle's train step on the host cores; baseline only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F16_PEAK_TFLOPS = 2500.0     # MI355X dense f16/bf16 MFMA (MI355X_MICROARCH.md)
TRAIN_GFLOP_PER_IMG = 516.5       # BASELINE.md §2, VGG-16 + PixelLink heads at 512x512


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(size, host_cpus, budget_s=60.0, runs=5, warmups=2):
    """The oracle's (CPU restatement, f32) full train step — forward + dice loss + backward — on a bounded sample: a
    batch of B images of the benchmark's size, B = the largest batch (<= 32, SURVEY 8d's) whose `warmups` + `runs`
    passes fit `budget_s` seconds of host time, sized from a probe pass at batch 2; MEDIAN of `runs` after `warmups`
    warm-ups.  Threads: all host CPUs up to 64 (small-batch oneDNN convolutions stop scaling before that)."""
    from oracle import ocr_oracle as O
    threads = max(1, min(host_cpus, 64))
    torch.set_num_threads(threads)
    rng = np.random.default_rng(0)
    p = O.init_model_vgg_params(rng)

    def one_pass(data):
        images, pixel, link, mask = data
        t0 = time.time()
        tp = O.to_torch_params(p)
        px, lk, _ = O.model_vgg(torch.from_numpy(images), tp, True, mixed=False)
        L = O.dice_loss(torch.from_numpy(pixel), px, torch.from_numpy(link), lk, torch.from_numpy(mask))
        L.backward()
        return time.time() - t0
    probe = one_pass(O.synthetic_batch(np.random.default_rng(1), 2, size))       # also pages oneDNN in
    per_img = probe / 2.0
    batch = int(budget_s / ((runs + warmups) * per_img))
    batch = max(2, min(32, batch - batch % 2))
    data = O.synthetic_batch(rng, batch, size)
    times = [one_pass(data) for _ in range(warmups + runs)]
    med = float(np.median(times[warmups:]))
    return {"value": round(batch / med, 4), "unit": "images/sec", "cores": threads, "host_cpus": host_cpus,
            "cpu_model": _cpu_model(), "kind": "port", "batch": batch,
            "sample": "batch of %d %dx%d images (the largest batch <= 32 whose %d passes fit a %.0f s budget, sized from "
                      "a probe pass at batch 2: %.1f s), forward+loss+backward, median of %d after %d warm-ups, torch-CPU "
                      "f32 oracle on %d threads (%.1f s of CPU work in all)"
                      % (batch, size, size, runs + warmups, budget_s, probe, runs, warmups, threads, probe + sum(times))}


CONFIG_LEGS = (
    # (key, scripts/bench_configs.py --which, environment of the child, BASELINE.json configs index)
    ("pixellink_vgg16_512_b32_train_decode", "pixellink", {}, 2),
    ("east_resnet50_640_b64_bf16_one_gpu_share", "resnet", {"OCR_STORAGE": "bf16"}, 3),
    ("pixellink_infer_1024_b16_decode_lanms", "decode", {}, 4),
    # the headline step WITH its input path (multigpu_train.py:164-174 + datasets/icdar.py:542-668): fed vs resident.
    # 128 synthetic JPEGs (20 MB) cycled: every timed read is served by the page cache — the leg measures decode + parse +
    # upload + resize / label kernels, NOT storage
    ("pipeline_train_vgg16_512_b32_fed_from_files_page_cached", "pipeline_train", {}, 1),
    # the precision that meets north_star's "score maps within 1e-3": f32 on the matrix cores, inference forward
    ("f32_inference_forward_model_vgg_512_pixellink_1024", "f32_forward", {}, 1),
)


def config_legs(steps=5, warmup=2, timeout=240):
    """Short legs of the OTHER BASELINE.json configs, after the timed region, N = 1 only.
    Each leg is a FRESH child process (`subprocess.run`, never exec; the bf16 leg needs its own interpreter
    because the storage type is process wide) running scripts/bench_configs.py; its one JSON line is
    condensed here.  A leg that fails or times out is reported as {"error": ...} — it never fails the bench."""
    import subprocess
    out = {}
    for key, which, extra_env, idx in CONFIG_LEGS:
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT",
                                                                  "OCR_STORAGE", "OCR_HIP_LIB")}
        env.update(extra_env)
        cmd = [sys.executable, os.path.join(ROOT, "scripts", "bench_configs.py"), "--which", which,
               "--steps", str(steps), "--warmup", str(warmup)]
        t0 = time.time()
        try:
            r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
            lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
            if r.returncode != 0 or not lines:
                out[key] = {"baseline_config": idx, "error": (r.stderr.decode()[-300:] or "no output")}
                continue
            j = json.loads(lines[-1])
            leg = {"baseline_config": idx, "workload": j.get("config"), "dtype": j.get("dtype", "f16"),
                   "steps": j.get("steps"), "ms_per_step": j.get("ms_per_step"), "img_s": j.get("images_per_sec"),
                   "tflops": j.get("tflops"), "frac_of_peak": j.get("frac_of_peak"),
                   "dominant_kernel": j.get("dominant_kernel"), "leg_wall_s": round(time.time() - t0, 1)}
            for k in ("loss", "decode_ms_per_batch", "net_forward_ms", "lanms_ms_per_batch", "lanms_boxes_per_sec",
                      "decode_algorithmic_GBps", "components", "peak_hbm_gib", "step_ms_fed", "step_ms_resident",
                      "fed_over_resident", "host_decode_img_s", "host_workers", "host_worker_kind", "host_cpus", "host_wait_ms_per_step",
                      "limiting_stage", "files_in_page_cache", "caveat"):
                if k in j:
                    leg[k] = j[k]
            out[key] = leg
        except Exception as e:                                        # timeout, bad JSON: report, do not fail
            out[key] = {"baseline_config": idx, "error": repr(e)[:300]}
    return out


def counters_from_profiles(dom):
    """`roofline.traffic / mfma_busy / clock_ghz` come from separate rocprofv3 --pmc passes and the diagnostic
    build (they cannot be collected inside an un-profiled run): committed summaries under profiles/.  Each file
    carries the fingerprint of the kernel sources it was measured on (`_provenance.csrc_sha16`,
    _lib.csrc_fingerprint); a field whose file is absent or was measured on other sources is null, and
    `counters_from` says which file, which fingerprint and when."""
    from tensorflow_ocr_amd import _lib
    now = _lib.csrc_fingerprint()

    def load(fname):
        try:
            with open(os.path.join(ROOT, "profiles", fname)) as f:
                return json.load(f)
        except Exception:
            return None
    src = {"csrc_sha16_now": now, "files": {}}
    vals = {"traffic": None, "mfma_busy": None, "clock_ghz": None}
    for field, fname in (("traffic", PROFILE_ROUND + "_pmc_traffic.json"), ("mfma_busy", PROFILE_ROUND + "_pmc_mfma.json"),
                         ("clock_ghz", PROFILE_ROUND + "_clock_diag.json")):
        j = load(fname)
        prov = (j or {}).get("_provenance") or {}
        fresh = j is not None and prov.get("csrc_sha16") == now
        src["files"][field] = {"file": "profiles/" + fname if j is not None else None,
                               "csrc_sha16": prov.get("csrc_sha16"), "date": prov.get("date"), "current": bool(fresh)}
        if not fresh:
            continue
        if field == "traffic":
            vals[field] = (j.get(dom) or {}).get("hbm_bytes_per_launch")
        elif field == "mfma_busy":
            vals[field] = (j.get(dom) or {}).get("mfma_busy_frac")
            vals["mfma_busy_clock_ghz"] = (j.get(dom) or {}).get("clock_ghz")
        else:
            ck = [v["clock_ghz_median"] for k, v in j.items() if k != "_provenance" and v.get("kernel") == dom
                  and v.get("operands") == "random"]
            vals[field] = round(sum(ck) / len(ck), 3) if ck else None
    return vals, src


PROFILE_ROUND = "r06"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU (reference: batch_size_per_gpu)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--loss-trace", action="store_true", help="also report the loss of EVERY step taken (the three "
                    "engine-build steps, the warm-up and the timed steps are all real optimiser steps on the same batch): "
                    "one host sync per step, so for checking (tests/test_gpu_batch_parity.py), not for timing")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the short legs of BASELINE configs[2..4]")
    ap.add_argument("--no-pg", action="store_true", help="N=1: do not run the bucketed exchange through a "
                    "one-rank RCCL communicator (by default it runs, so the line records that RCCL loaded)")
    ap.add_argument("--no-proxy", action="store_true", help="N=1: skip the comm-kernel stand-in legs (profiling passes: their "
                    "stretched steps would enter the per-kernel averages)")
    ap.add_argument("--proxy-workgroups", type=int, default=24, help="N=1: workgroups of the comm-kernel stand-in (RCCL runs "
                    "16-32 channels)")
    ap.add_argument("--proxy-link-gbps", type=float, default=150.0, help="N=1: pace of the stand-in (one xGMI link)")
    ap.add_argument("--loss-scale", type=float, default=1024.0)
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL); "
                    "gloo lets several ranks share one GPU for a functional check")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (functional check only)")
    ap.add_argument("--force-pg", action="store_true", help="N=1 only: create a one-rank process group and run "
                    "the bucketed gradient exchange anyway (RCCL stream ordering on a one-GPU box)")
    args = ap.parse_args()

    # Plain `python bench.py --gpus N`: this process becomes the launcher of N ranks (one per GPU) and
    # relays rank 0's JSON line; it returns here only as a rank (WORLD_SIZE set) or for N = 1.
    # Nothing above this line has touched HIP.
    from tensorflow_ocr_amd import launch
    rc = launch.self_launch(args.gpus)
    if rc is not None:
        sys.exit(rc)

    # The contract is ONE JSON line on stdout.  Libraries print there too (RCCL's five-line version banner when a
    # communicator comes up, gloo's connection notes): from here on file descriptor 1 points at stderr, and the line
    # at the end is written to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    from tensorflow_ocr_amd import _lib, dist, ops, synthetic
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import model_vgg_16 as M
    from tensorflow_ocr_amd.train import AdamOptimizer, TrainStep
    import torch.distributed as td

    if args.share_gpu:
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local = dist.init_process_group_from_env(args.backend, force=args.force_pg)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    # N = 1: a single tower exchanges nothing (the reference's tower loop has one tower), so the TIMED steps run without
    # an exchange.  The recorded step still CONTAINS the bucketed exchange through a ONE-rank RCCL communicator (C ABI:
    # ocr_allreduce_bucket on a comm stream, event-ordered against the compute stream) — switched off for the timed
    # region and on for the A/B after it, so the single-GPU line shows that RCCL loaded and what the exchange path
    # costs (`exchange.ms_per_step_with_exchange`).  Falls back to no exchange if RCCL cannot be initialised here.
    force_reduce = args.force_pg
    exchange_error = None
    if world == 1 and not args.no_pg and not args.force_pg:
        try:
            if dist.exchange_mode(1, True, force=True) == "abi":
                dist.AbiComm(0, 1).destroy()          # probe: RCCL present and a one-rank communicator comes up
                force_reduce = True
        except Exception as e:
            exchange_error = repr(e)[:200]

    g = Graph(device, loss_scale=args.loss_scale, seed=1)           # same init on every rank
    rng = np.random.default_rng(100 + rank)                          # different data per rank
    images, pixel, link, mask = synthetic.make_batch(rng, args.batch, args.size)
    batch = [torch.from_numpy(a).to(device) for a in (images, pixel, link, mask)]   # resident in HBM

    def forward_loss(gr, im, px, lk, mk):
        f_score, f_geometry = M.model_vgg(im, is_training=True, graph=gr)
        return M.loss(px, f_score, lk, f_geometry, mk, graph=gr)

    # N = 1 with the one-rank exchange recorded: the plan also holds, next to every bucket's all-reduce, a stand-in with
    # the shape of a multi-rank ring's device code (ocr_comm_proxy) for the `exchange.proxy` leg below
    proxy_cfg = (args.proxy_workgroups, args.proxy_link_gbps) if (world == 1 and force_reduce and not args.force_pg and not args.no_proxy) else None
    step = TrainStep(g, forward_loss, lambda gr: AdamOptimizer(gr, learning_rate=1e-4), world_size=world,
                     force_reduce=force_reduce, comm_proxy=proxy_cfg)

    def barrier():
        if td.is_initialized():
            td.barrier()
        torch.cuda.synchronize()

    loss = None
    trace = [] if args.loss_trace else None

    def take(l):
        if trace is not None:
            trace.append(round(float(l.item()), 6))
        return l
    for _ in range(3):          # engine build: variables, flat buffers, recorded step plan (untimed)
        loss = take(step(*batch))
    probe_only = world == 1 and force_reduce and not args.force_pg
    if probe_only:
        step.reducer.enabled = False          # the plan holds the exchange entries; a single tower does not run them
    for _ in range(args.warmup):
        loss = take(step(*batch))
    barrier()
    ops.KERNEL_TIMING = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = take(step(*batch))
    barrier()
    dt = time.perf_counter() - t0
    timing, ops.KERNEL_TIMING = ops.KERNEL_TIMING, None

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=device)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        return float(t.item())
    dt = max_over_ranks(dt)
    loss_val = loss.item()

    # data-parallel invariants, checked before the A/B below lets the towers drift: every rank trained on its OWN shard
    # (checksums of the input images all differ) and every rank holds the SAME parameters (identical initialisation, the
    # same averaged gradient every step — multigpu_train.py:70-85)
    replicas = None
    if world > 1:
        mine = torch.tensor([float(g.store.flat.double().sum().item()), float(batch[0].double().sum().item())],
                            dtype=torch.float64, device=device)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        td.all_gather(allv, mine)
        par = [float(v[0].item()) for v in allv]
        dat = [float(v[1].item()) for v in allv]
        replicas = {"parameters_identical": len(set(par)) == 1, "data_shards_distinct": len(set(dat)) == world,
                    "ranks": world}

    # the exchange itself, outside the timed region: (1) a collective that proves every rank is in the
    # RCCL communicator, (2) the same K steps with the gradient exchange switched off — the difference
    # is the communication the backward pass did NOT hide.  (The towers' weights drift apart in (2);
    # nothing is measured after it.)
    comm = None
    red = step.reducer
    if red is not None and red.active:
        if red.mode == "abi":
            ones = torch.ones(1, dtype=torch.float32, device=device)
            red.comm.all_reduce_(ones)                     # through ocr_allreduce_bucket, on the compute stream
            backend = "rccl (C ABI: ocr_allreduce_bucket)"
        else:
            ones = torch.ones(1, dtype=torch.float32, device=device)
            td.all_reduce(ones)
            backend = td.get_backend()
        red.enabled = probe_only              # N = 1: the A/B leg is the one WITH the exchange; N > 1: the one without
        for _ in range(2):
            step(*batch)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(*batch)
        barrier()
        dt_ab = max_over_ranks(time.perf_counter() - t1)
        red.enabled = not probe_only
        comm = {"backend": backend, "mode": red.mode, "rccl_ranks": int(round(ones.item())),
                "bucket_bytes": red.bucket_nbytes(), "grad_bytes": int(g.store.flat_grad.numel() * 4)}
        if probe_only:
            comm["in_timed_region"] = False
            comm["ms_per_step_with_exchange"] = round(dt_ab / args.steps * 1e3, 3)
            comm["comm_exposed_ms"] = round((dt_ab - dt) / args.steps * 1e3, 3)
        else:
            comm["in_timed_region"] = True
            comm["ms_per_step_no_exchange"] = round(dt_ab / args.steps * 1e3, 3)
            comm["comm_exposed_ms"] = round((dt - dt_ab) / args.steps * 1e3, 3)
    elif exchange_error is not None:
        comm = {"backend": None, "error": exchange_error}
    # One-GPU proxy for the comm / compute overlap: the same K steps (a) without any exchange and (b)
    # with ocr_comm_proxy on the comm stream in place of every bucket's all-reduce, A/B/A/B interleaved; the end of
    # backward on the compute stream is marked with an event, so `backward_stretch_ms` is what the stand-in costs the
    # conv workgroups it shares the chip with and `step_ms_with - step_ms_without` what stays exposed.
    if comm is not None and red is not None and red.active and red.mode == "abi" and red.proxy is not None and probe_only:
        # the fat arms (round 6): the same stand-in with ~128 VGPRs + 64 KB of LDS per workgroup on 32 workgroups — with the
        # 18-register one they bracket what a real ring's kernels cost the step (ocr_comm_proxy_set_footprint)
        from tensorflow_ocr_amd import _lib as _L
        # "at_fork" arms (item 7c): a completed bucket's stand-in is issued at the NEXT fork of the recorded step, beside held-back
        # weight gradients, instead of right behind the join (train.schedule_guests(xchg_at_fork=True))
        arms = (("without", False, True, 0, False), ("overlapped", True, True, 0, False), ("after_backward", True, False, 0, False),
                ("overlapped_fat", True, True, 1, False), ("after_backward_fat", True, False, 1, False),
                ("overlapped_at_fork", True, True, 0, True), ("overlapped_fat_at_fork", True, True, 1, True))
        res = {a[0]: [[], []] for a in arms}
        for rnd in range(2):
            for arm, with_proxy, overlap, fat, at_fork in arms:
                _L.call("ocr_comm_proxy_set_footprint", ctypes.c_int(fat), ctypes.c_int(32 if fat else 0))
                step.reschedule(xchg_at_fork=at_fork)
                red.enabled, red.use_proxy, red.overlap = with_proxy, with_proxy, overlap
                step.backward_end_event = None
                step(*batch)
                barrier()
                evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
                t1 = time.perf_counter()
                for e0, e1 in evs:                       # no host sync inside the loop: the events are read afterwards
                    e0.record()
                    step.backward_end_event = e1
                    step(*batch)
                barrier()
                res[arm][0].append((time.perf_counter() - t1) / args.steps * 1e3)
                res[arm][1].append(float(np.median([e0.elapsed_time(e1) for e0, e1 in evs])))
        stats = red.proxy_stats.cpu().numpy()
        _L.call("ocr_comm_proxy_set_footprint", ctypes.c_int(0), ctypes.c_int(0))
        step.reschedule()
        red.enabled, red.use_proxy, red.overlap = False, False, True
        step.backward_end_event = None
        nb = len(red.buckets)
        busy_ms = float(stats[3]) / 1e5 / max(int(stats[4]), 1) * nb          # 100 MHz ticks -> ms, per step
        paced_ms = 2.0 * comm["grad_bytes"] / (red.proxy[1] * 1e9) * 1e3
        best = {k: (min(v[0]), min(v[1])) for k, v in res.items()}
        comm["proxy"] = {
            "kernel": "ocr_comm_proxy: %d workgroups x 256 threads on the comm stream, one launch per bucket (same events as "
                      "the all-reduce), 2 x bucket bytes through HBM in 256 KB chunks paced at %.0f GB/s" % red.proxy,
            "step_ms_without": round(best["without"][0], 3),
            "step_ms_with": round(best["overlapped"][0], 3),
            "step_ms_with_after_backward": round(best["after_backward"][0], 3),
            "backward_ms_without": round(best["without"][1], 3), "backward_ms_with": round(best["overlapped"][1], 3),
            "backward_ms_with_after_backward": round(best["after_backward"][1], 3),
            "backward_stretch_ms": round(best["overlapped"][1] - best["without"][1], 3),
            "exposed_ms_overlapped": round(best["overlapped"][0] - best["without"][0], 3),
            "exposed_ms_after_backward": round(best["after_backward"][0] - best["without"][0], 3),
            "proxy_busy_ms_per_step": round(busy_ms, 3), "proxy_paced_ms_per_step": round(paced_ms, 3),
            "proxy_launches": int(stats[4]), "rounds": "three arms interleaved twice, best of 2 per arm",
            "reading": "overlapped: the buckets' stand-ins run under backward (dist.GradientAllReduce(overlap=True)); "
                       "after_backward: all of them between backward and the optimiser (overlap=False); DESIGN.md 3.4"}
        comm["proxy_fat"] = {
            "kernel": "the same stand-in at ~128 VGPRs + 64 KB of LDS per workgroup, 32 workgroups (RCCL-like footprint)",
            "step_ms_with": round(best["overlapped_fat"][0], 3), "step_ms_with_after_backward": round(best["after_backward_fat"][0], 3),
            "backward_stretch_ms": round(best["overlapped_fat"][1] - best["without"][1], 3),
            "exposed_ms_overlapped": round(best["overlapped_fat"][0] - best["without"][0], 3),
            "exposed_ms_after_backward": round(best["after_backward_fat"][0] - best["without"][0], 3)}
        comm["proxy_at_fork"] = {
            "what": "the buckets' stand-ins issued at the next fork of the recorded step (train.schedule_guests(xchg_at_fork=True)) "
                    "instead of right behind the join: beside held-back weight gradients, the only kernels they can share a CU with",
            "exposed_ms_overlapped": round(best["overlapped_at_fork"][0] - best["without"][0], 3),
            "backward_stretch_ms": round(best["overlapped_at_fork"][1] - best["without"][1], 3),
            "exposed_ms_overlapped_fat": round(best["overlapped_fat_at_fork"][0] - best["without"][0], 3),
            "backward_stretch_ms_fat": round(best["overlapped_fat_at_fork"][1] - best["without"][1], 3)}
        # the placement this build ships and what each placement projects to at 8 ranks, from THIS run's
        # numbers (a projection from a one-GPU stand-in paced at one xGMI link: RCCL with more than one rank has never run here)
        pr = comm["proxy"]
        w0 = pr["step_ms_without"]
        comm["plan"] = {
            "placement": "under_backward" if pr["exposed_ms_overlapped"] <= pr["exposed_ms_after_backward"] else "after_backward",
            "default_placement": "under_backward (dist.GradientAllReduce(overlap=True); OCR_EXCHANGE_OVERLAP=0 places every bucket after backward)",
            "buckets": len(red.buckets), "bucket_bytes": comm.get("bucket_bytes"), "rccl_ranks_in_this_run": 1,
            "projected_step_ms_8_ranks": {"under_backward": round(w0 + pr["exposed_ms_overlapped"], 3),
                                          "after_backward": round(w0 + pr["exposed_ms_after_backward"], 3)},
            "projected_scaling_of_8": {"under_backward": round(8.0 * w0 / (w0 + pr["exposed_ms_overlapped"]), 2),
                                       "after_backward": round(8.0 * w0 / (w0 + pr["exposed_ms_after_backward"]), 2)},
            # the bracket: the 18-register stand-in is the optimistic end, the RCCL-like footprint the other
            "projected_scaling_of_8_bracket": {
                "under_backward": [round(8.0 * w0 / (w0 + max(comm["proxy_fat"]["exposed_ms_overlapped"], 0.0)), 2),
                                   round(8.0 * w0 / (w0 + max(pr["exposed_ms_overlapped"], 0.0)), 2)],
                "after_backward": [round(8.0 * w0 / (w0 + max(comm["proxy_fat"]["exposed_ms_after_backward"], 0.0)), 2),
                                   round(8.0 * w0 / (w0 + max(pr["exposed_ms_after_backward"], 0.0)), 2)]},
            "why_the_stand_in_is_not_hidden": "its workgroups (18 registers) are placed beside the weight gradients and the 64-channel "
                                              "kernel, but a 512-register input-gradient wave cannot share a CU with them: those launches "
                                              "run one round longer (profiles/r06_coresidency_probe.json; DESIGN.md 3.5a)"}
    # dominant kernel: the conv_igemm instantiation with the most accumulated time
    per, fwd = {}, {}
    for variant, flops, phase, e0, e1 in timing:
        sec = e0.elapsed_time(e1) * 1e-3
        a = per.setdefault(variant, [0.0, 0.0, 0])
        a[0] += flops
        a[1] += sec
        a[2] += 1
        if phase == "fwd":          # forward launches
            b = fwd.setdefault(variant, [0.0, 0.0])
            b[0] += flops
            b[1] += sec
    dom = max(per, key=lambda k: per[k][1]) if per else None
    roof = None
    if dom:
        fl, sec, cnt = per[dom]
        ach = fl / sec / 1e12
        vals, counters_from = counters_from_profiles(dom)
        traffic, mfma_busy, clock = vals["traffic"], vals["mfma_busy"], vals["clock_ghz"]
        # `achieved` averages EVERY launch of the kernel in the timed region (forward and input-gradient
        # launches), as rocprofv3 --stats does; `achieved_alone` = the forward launches only (the
        # input-gradient ones also carry the fused BN-backward sums in their epilogue).
        alone = round(fwd[dom][0] / fwd[dom][1] / 1e12, 1) if dom in fwd and fwd[dom][1] > 0 else None
        roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 1), "peak": MFMA_F16_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(ach / MFMA_F16_PEAK_TFLOPS, 4), "traffic": traffic,
                "launches_per_step": cnt // args.steps, "avg_launch_ms": round(sec / cnt * 1e3, 4),
                "share_of_step": round(sec / dt, 3), "achieved_alone": alone,
                "frac_alone": round(alone / MFMA_F16_PEAK_TFLOPS, 4) if alone else None,
                "clock_ghz": clock, "mfma_busy": mfma_busy, "mfma_busy_clock_ghz": vals.get("mfma_busy_clock_ghz"),
                # which clock goes with which counter: mfma_busy is a CYCLE ratio of the PMC pass and
                # pairs with that pass's own clock (GRBM_GUI_ACTIVE / launch time: mfma_busy_clock_ghz) — frac ~ mfma_busy x
                # mfma_busy_clock_ghz / 2.4 x (MFMA issue slots filled while busy); clock_ghz is the diagnostic build's
                # in-kernel s_memrealtime clock of the main loop alone (no prologue / epilogue), a different run
                "clock_note": "mfma_busy pairs with mfma_busy_clock_ghz (same PMC pass); clock_ghz = in-kernel main-loop clock of the diagnostic build",
                "counters_from": counters_from}
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * args.batch * args.steps / dt
        out = {
            "metric": "images/sec training, 512x512 ICDAR, VGG-16 EAST, batch 32, 1/2/4/8 GPU",
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": _lib.STORAGE, "data": "synthetic",
            "config": {"workload": "VGG-16 model_vgg + dice loss train step, %dx%d, batch %d per GPU, "
                                   "%s storage / f32 accumulate, Adam+EMA" % (args.size, args.size, args.batch, _lib.STORAGE),
                       "global_batch": world * args.batch, "parallelism": "dp%d" % world,
                       "loss_scale": args.loss_scale},
            "loss": round(loss_val, 5),
            "train_tflops": round(value * TRAIN_GFLOP_PER_IMG / 1e3 * (args.size / 512.0) ** 2, 1),
            "roofline": roof,
        }
        if trace is not None:
            out["loss_trace"] = trace
        if replicas is not None:
            out["replicas"] = replicas
        if comm is not None:
            out["exchange"] = comm
        if getattr(step, "plan", None) is not None:
            # train.schedule_guests: how much of the recorded step runs as guest / host pairs
            kinds = [e[0] if e[0] != "c" else (e[4][0] if e[4] is not None else "c") for e in step.plan]
            paired = [e for e in step.plan if e[0] == "c" and e[4] is not None and e[4][0] == "guest" and e[4][-1] == "paired"]
            out["guests"] = {"pairs": kinds.count("fork"), "guest_passes": sum(1 for k in kinds if k == "guest"),
                             "weight_gradients": sum(1 for e in step.plan if e[0] == "c" and e[4] is not None and e[4][0] == "side" and len(e[4]) > 1),
                             "guest_bytes_per_step_GB": round(sum(e[4][1] for e in paired) / 1e9, 2),
                             "second_stream": "guest passes (<= 56 registers per lane) beside held-back weight-gradient slab kernels"}
        if world == 1 and not args.no_config_legs:
            # free this process's activations first: the legs are children that need the HBM
            del step, batch
            g.tape.clear()
            torch.cuda.empty_cache()
            out["configs"] = config_legs()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.size, os.cpu_count() or 1)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if td.is_available() and td.is_initialized():
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
