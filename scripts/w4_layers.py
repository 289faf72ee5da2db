#!/usr/bin/env python3
"""Per-layer table of the dominant kernel (`conv3x3_w4_kernel`) inside the headline step.

    python3 scripts/w4_layers.py <kernel-trace dir> <pmc_mfma dir> > profiles/rNN_w4_per_layer.json

Inputs: the `rocprofv3 --kernel-trace --stats` run of bench.py (launch durations) and the SQ counter pass of the same
command (scripts/pmc_mfma.py's input: SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_WAIT_* per dispatch).  The kernel
is launched 17 times per step, always in the same order: the forward convolutions conv3_1 ... conv5_3 (9), then the
input gradients conv5_3 ... conv3_2 (8, each with the fused BN-backward reduction of the layer below in its epilogue);
launch i of a step is therefore layer i of the list below.  Steps are cut at the optimiser launch; the first three
(engine build) are dropped.  FLOPs per launch = 2 * N * H * W * Cin * Cout * 9 at batch 32."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_mfma import provenance  # noqa: E402

N = 32
FWD = [("conv3_1", 128, 128, 256), ("conv3_2", 128, 256, 256), ("conv3_3", 128, 256, 256),
       ("conv4_1", 64, 256, 512), ("conv4_2", 64, 512, 512), ("conv4_3", 64, 512, 512),
       ("conv5_1", 32, 512, 512), ("conv5_2", 32, 512, 512), ("conv5_3", 32, 512, 512)]
# input-gradient launches, in backward order; conv3_1's (256 -> 128 couts) and conv4_1's (512 -> 256: cin of the
# gradient GEMM = 512) run this kernel only when the GRADIENT's cout (= the layer's cin) is a multiple of 256
BWD = [(n + " dgrad", hw, co, ci) for n, hw, ci, co in reversed(FWD) if ci % 256 == 0]
LAYERS = [(n, hw, ci, co, "fwd") for n, hw, ci, co in FWD] + [(n, hw, ci, co, "dgrad") for n, hw, ci, co in BWD]
PEAK = 2500e12


def per_step(rows, kernel, marker="adam_kernel", skip=3):
    rows = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
    steps, cur, seen = [], [], 0
    for r in rows:
        if kernel in r["Kernel_Name"]:
            cur.append(r)
        if marker in r["Kernel_Name"]:
            seen += 1
            if seen > skip and cur:
                steps.append(cur)
            cur = []
    return steps


def main(trace_dir, pmc_dir):
    tr = list(csv.DictReader(open(glob.glob(trace_dir + "/**/*kernel_trace.csv", recursive=True)[0])))
    steps = [s for s in per_step(tr, "conv3x3_w4_kernel") if len(s) == len(LAYERS)]
    if not steps:
        raise SystemExit("no step with %d launches of the kernel" % len(LAYERS))
    # PMC pass: counters per dispatch, joined with ITS OWN kernel trace for the order
    ptr = list(csv.DictReader(open(glob.glob(pmc_dir + "/**/*kernel_trace.csv", recursive=True)[0])))
    cnt = collections.defaultdict(dict)
    for r in csv.DictReader(open(glob.glob(pmc_dir + "/**/*counter_collection.csv", recursive=True)[0])):
        cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    psteps = [s for s in per_step(ptr, "conv3x3_w4_kernel") if len(s) == len(LAYERS)]
    out = {"_provenance": provenance("scripts/w4_layers.py on the --kernel-trace run and the SQ --pmc pass of bench.py"),
           "steps_averaged": len(steps), "pmc_steps_averaged": len(psteps), "layers": []}
    tot_us = tot_fl = 0.0
    for i, (name, hw, ci, co, phase) in enumerate(LAYERS):
        us = sum((int(s[i]["End_Timestamp"]) - int(s[i]["Start_Timestamp"])) for s in steps) / len(steps) / 1e3
        flops = 2.0 * N * hw * hw * ci * co * 9
        row = {"launch": i, "layer": name, "phase": phase, "hw": hw, "cin": ci, "cout": co,
               "workgroups": N * (hw // 8) * (hw // 32) * (co // 256), "avg_us": round(us, 1),
               "tflops": round(flops / us / 1e6, 1), "frac_of_peak": round(flops / us / 1e6 / (PEAK / 1e12), 4),
               "mfma_ideal_us": round(flops / PEAK * 1e6, 1)}
        if psteps:
            def mean(key):
                vals = [cnt[s[i]["Dispatch_Id"]].get(key, 0.0) for s in psteps]
                return sum(vals) / len(vals)
            cyc = mean("GRBM_GUI_ACTIVE") / 8.0
            wc = max(mean("SQ_WAVE_CYCLES"), 1.0)
            pus = sum((int(s[i]["End_Timestamp"]) - int(s[i]["Start_Timestamp"])) for s in psteps) / len(psteps) / 1e3
            row.update({"kernel_cycles": round(cyc), "mfma_busy_frac": round(mean("SQ_VALU_MFMA_BUSY_CYCLES") / 1024.0 / max(cyc, 1.0), 3),
                        "wave_parked_frac": round(mean("SQ_WAIT_ANY") / wc, 3),
                        "issue_stall_frac": round(mean("SQ_WAIT_INST_ANY") / wc, 3),
                        "clock_ghz_in_pmc_pass": round(cyc / (pus * 1e3), 3)})
        out["layers"].append(row)
        tot_us += us
        tot_fl += flops
    out["total"] = {"us_per_step": round(tot_us, 1), "tflops": round(tot_fl / tot_us / 1e6, 1),
                    "frac_of_peak": round(tot_fl / tot_us / 1e6 / (PEAK / 1e12), 4)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
