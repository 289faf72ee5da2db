"""Kernel sequence of the LAST step of a rocprofv3 --kernel-trace run of a training command.

    python3 scripts/step_trace.py <trace_dir> <dispatches_per_step> [--agg]

The step length is what `make_profiles.sh` prints ("N dispatches after step 3, K steps" -> N / K)."""
import collections
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_mfma import short  # noqa: E402

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
per = int(sys.argv[2])
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
last = rows[-per:]
tot = 0.0
agg = collections.Counter()
cnt = collections.Counter()
for i, r in enumerate(last):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    agg[short(r["Kernel_Name"])] += d
    cnt[short(r["Kernel_Name"])] += 1
    if "--agg" not in sys.argv:
        print(i, short(r["Kernel_Name"])[:60], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"], round(d, 1))
print("total_us", round(tot, 1), "dispatches", len(last))
if "--agg" in sys.argv:
    for k, v in agg.most_common():
        print(round(v, 1), cnt[k], k)
