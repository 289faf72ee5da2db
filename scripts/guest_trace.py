"""Guest / host pairs in a rocprofv3 --kernel-trace run of bench.py (train.schedule_guests): for the last complete step,
every guest kernel (csrc/guest_bn.hip) with the weight-gradient kernel it ran beside — start offsets, durations, the gap
to the next main-stream kernel — and the totals.

    python3 scripts/guest_trace.py <trace_dir> [--json]"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_mfma import short  # noqa: E402


def main():
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    ks = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "")) for r in rows]
    # last complete step: between the last two adam kernels
    adam = [i for i, k in enumerate(ks) if "adam_kernel" in k[0]]
    lo, hi = adam[-2] + 1, adam[-1] + 1
    step = ks[lo:hi]
    t0 = step[0][1]
    span = (step[-1][2] - t0) / 1e3
    busy = sum((k[2] - k[1]) for k in step) / 1e3
    pairs = []
    for i, k in enumerate(step):
        if "affine" not in k[0]:
            continue
        # the host: the wgrad kernel whose interval intersects this guest's the most
        best, ov = None, 0
        for h in step:
            if "wgrad" in h[0] and "slab" not in h[0]:
                o = min(h[2], k[2]) - max(h[1], k[1])
                if o > ov:
                    best, ov = h, o
        nxt = [h for h in step if h[1] >= max(k[2], best[2] if best else k[2]) - 2000 and h is not k and h is not best and "slab_reduce" not in h[0]]
        pairs.append({"guest": k[0][:40], "guest_us": round((k[2] - k[1]) / 1e3, 1), "guest_start_us": round((k[1] - t0) / 1e3, 1),
                      "host": best[0][:32] if best else None, "host_us": round((best[2] - best[1]) / 1e3, 1) if best else None,
                      "guest_start_after_host_start_us": round((k[1] - best[1]) / 1e3, 1) if best else None,
                      "overlap_us": round(ov / 1e3, 1),
                      "guest_end_after_host_end_us": round((k[2] - best[2]) / 1e3, 1) if best else None,
                      "queue_guest": k[3], "queue_host": best[3] if best else None})
    out = {"step_span_us": round(span, 1), "sum_of_kernel_us": round(busy, 1), "dispatches": len(step), "pairs": pairs}
    if "--json" in sys.argv:
        print(json.dumps(out, indent=1))
        return
    print("step span %.1f us, sum of kernel time %.1f us, %d dispatches" % (span, busy, len(step)))
    for p in pairs:
        print("%-42s %7.1f us @%8.1f | host %-30s %7.1f us | guest starts +%6.1f, overlap %7.1f, guest ends %+7.1f vs host end  q %s/%s" % (
            p["guest"], p["guest_us"], p["guest_start_us"], p["host"], p["host_us"] or 0, p["guest_start_after_host_start_us"] or 0,
            p["overlap_us"], p["guest_end_after_host_end_us"] or 0, p["queue_guest"], p["queue_host"]))
    if "--dump" in sys.argv:
        for k in step:
            print("%9.1f %9.1f %8.1f q%s %s" % ((k[1] - t0) / 1e3, (k[2] - t0) / 1e3, (k[2] - k[1]) / 1e3, k[3], k[0][:70]))


if __name__ == "__main__":
    main()
