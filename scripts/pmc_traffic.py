"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel.

    python3 scripts/pmc_traffic.py <fetch_dir> <write_dir> > profiles/rNN_pmc_traffic.json

Per MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a
wide (16 B/lane) coalesced read stream -> doubled; WRITE_SIZE is exact for 16 B/lane stores.  Keys are the
kernel names as `ocr_conv2d_variant` / bench.py print them (scripts/pmc_mfma.py: short())."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_mfma import provenance, short  # noqa: E402

out = {}
for name, d in (("FETCH_SIZE", sys.argv[1]), ("WRITE_SIZE", sys.argv[2])):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(k, {})[name] = {"launches": len(v), "avg_KiB": sum(v) / len(v)}
res = {}
for k, v in out.items():
    fetch = v.get("FETCH_SIZE", {}).get("avg_KiB", 0.0) * 1024 * 2      # gfx950 correction
    write = v.get("WRITE_SIZE", {}).get("avg_KiB", 0.0) * 1024
    if fetch + write < 1 << 20:
        continue                                                        # tiny kernels: not worth a row
    res[k] = {"hbm_read_bytes_per_launch": round(fetch), "hbm_write_bytes_per_launch": round(write),
              "hbm_bytes_per_launch": round(fetch + write), "launches_profiled": v.get("FETCH_SIZE", {}).get("launches")}
cmd = sys.argv[3] if len(sys.argv) > 3 else "python3 bench.py"
final = {"_provenance": provenance("rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two passes) -- " + cmd)}
# whole-pass totals (every kernel, tiny ones included): sum over kernels of launches x average bytes
tot_r = sum(v.get("FETCH_SIZE", {}).get("avg_KiB", 0.0) * 2048 * v.get("FETCH_SIZE", {}).get("launches", 0) for v in out.values())
tot_w = sum(v.get("WRITE_SIZE", {}).get("avg_KiB", 0.0) * 1024 * v.get("WRITE_SIZE", {}).get("launches", 0) for v in out.values())
final["_totals"] = {"hbm_read_bytes_profiled_pass": round(tot_r), "hbm_write_bytes_profiled_pass": round(tot_w),
                    "note": "all launches of the profiled command (engine build + warm-up + timed steps): divide by the pass's step count"}
final.update(dict(sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * (kv[1]["launches_profiled"] or 1))))
print(json.dumps(final, indent=1))
