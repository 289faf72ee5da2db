#!/usr/bin/env python3
"""Gaps between consecutive dispatches of scripts/gap_probe/gap_probe.hip, grouped by (previous kernel, next kernel)."""
import collections, csv, glob, json, statistics, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
g = collections.defaultdict(list)
d = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    da = (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3
    g[("%s[~%d us]" % (a["Kernel_Name"].split("(")[0], 2 ** round(__import__("math").log2(max(da, 1.0)))), b["Kernel_Name"].split("(")[0])].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
for r in rows:
    d[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {"gap_us_median": {"%s -> %s" % k: round(statistics.median(v), 2) for k, v in g.items() if len(v) >= 10},
       "duration_us_median": {k: round(statistics.median(v), 2) for k, v in d.items()}}
json.dump(out, sys.stdout, indent=1)
