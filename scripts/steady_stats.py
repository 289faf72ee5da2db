#!/usr/bin/env python3
"""Steady-state per-kernel statistics from a rocprofv3 kernel trace.

    python3 scripts/steady_stats.py <rocprof output dir> [--skip-steps N] [--marker adam_kernel] > profiles/rNN_..._kernel_stats.csv

`rocprofv3 --kernel-trace --stats` averages EVERY dispatch of the process, including the first launch of each
kernel, which pays the lazy code-object load (29.9 ms for conv3x3_w4_kernel in the round-3 trace: +63 us on a
476-launch average of a 348 us kernel).  This script recomputes the same table (same columns as rocprofv3's
*_kernel_stats.csv) from *_kernel_trace.csv over the dispatches AFTER the first N training steps; a step ends at
each dispatch of the marker kernel (the optimiser launch: one per step).  Default N = 3 = the engine-build steps
of bench.py / scripts/bench_configs.py (eager execution, variable creation, plan recording)."""
import argparse
import csv
import glob
import math
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--skip-steps", type=int, default=3)
    ap.add_argument("--marker", default="adam_kernel", help="substring of the kernel launched once per step")
    args = ap.parse_args()
    f = glob.glob(args.dir + "/**/*kernel_trace.csv", recursive=True)
    if not f:
        raise SystemExit("no kernel_trace.csv under " + args.dir)
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    seen = 0
    start = 0
    for i, r in enumerate(rows):
        if args.marker in r["Kernel_Name"]:
            seen += 1
            if seen == args.skip_steps:
                start = i + 1
                break
    if seen < args.skip_steps:
        raise SystemExit("marker %r seen %d times, fewer than --skip-steps" % (args.marker, seen))
    per = {}
    for r in rows[start:]:
        per.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = float(sum(sum(v) for v in per.values()))
    w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        n, s = len(v), sum(v)
        mean = s / n
        sd = math.sqrt(sum((x - mean) ** 2 for x in v) / (n - 1)) if n > 1 else 0.0
        w.writerow([name, n, s, round(mean, 6), round(100.0 * s / total, 4), min(v), max(v), round(sd, 6)])
    steps = sum(1 for r in rows[start:] if args.marker in r["Kernel_Name"])
    sys.stderr.write("steady state: %d dispatches after step %d, %d steps, %.3f ms of kernel time per step\n" % (
        len(rows) - start, args.skip_steps, steps, total / max(steps, 1) / 1e6))


if __name__ == "__main__":
    main()
