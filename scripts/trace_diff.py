"""Two rocprofv3 --kernel-trace runs of bench.py (A, B): per-kernel-name time of the last complete step, side by side, and
the idle time of the main queue (gaps between consecutive kernels of the busiest queue).
    python3 scripts/trace_diff.py <dirA> <dirB>"""
import collections
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_mfma import short  # noqa: E402


def last_step(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    ks = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "")) for r in rows]
    adam = [i for i, k in enumerate(ks) if "adam_kernel" in k[0]]
    steps = []
    for a, b in zip(adam[-4:-1], adam[-3:]):
        steps.append(ks[a + 1:b + 1])
    return steps


def summarize(steps):
    agg = collections.Counter()
    cnt = collections.Counter()
    span = gaps = 0.0
    for st in steps:
        span += (st[-1][2] - st[0][1]) / 1e3
        qs = collections.Counter(k[3] for k in st)
        mainq = qs.most_common(1)[0][0]
        prev = None
        for k in st:
            agg[k[0]] += (k[2] - k[1]) / 1e3
            cnt[k[0]] += 1
            if k[3] == mainq:
                if prev is not None and k[1] > prev:
                    gaps += (k[1] - prev) / 1e3
                prev = max(prev or 0, k[2])
    n = len(steps)
    return {k: v / n for k, v in agg.items()}, {k: v / n for k, v in cnt.items()}, span / n, gaps / n


def main():
    A, ca, sa, ga = summarize(last_step(sys.argv[1]))
    B, cb, sb, gb = summarize(last_step(sys.argv[2]))
    print("step span us: A %.1f  B %.1f   main-queue idle us: A %.1f  B %.1f   sum of kernels: A %.1f B %.1f" % (
        sa, sb, ga, gb, sum(A.values()), sum(B.values())))
    for k in sorted(set(A) | set(B), key=lambda k: -abs(B.get(k, 0) - A.get(k, 0))):
        if abs(B.get(k, 0) - A.get(k, 0)) >= 3:
            print("%-46s A %5.1f x %8.1f | B %5.1f x %8.1f | %+8.1f" % (k[:46], ca.get(k, 0), A.get(k, 0), cb.get(k, 0), B.get(k, 0), B.get(k, 0) - A.get(k, 0)))


if __name__ == "__main__":
    main()
