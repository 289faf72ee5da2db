#!/usr/bin/env python3
"""Regenerates the tables of DESIGN.md section 5 that are read off profiles/rNN_*.{json,csv}, between the
`<!-- generated: NAME -->` / `<!-- end generated -->` markers, so that the text always shows the committed profiles.

    python3 scripts/design_tables.py r03
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from pmc_mfma import short  # noqa: E402


def w4_table(n):
    d = json.load(open(os.path.join(ROOT, "profiles", n + "_w4_per_layer.json")))
    out = ["| # | layer | workgroups | µs | TFLOP/s | of peak | MFMA busy | parked | issue-stalled |",
           "|---|---|---|---|---|---|---|---|---|"]
    for r in d["layers"]:
        out.append("| %d | %s | %d | %.1f | %.0f | %.3f | %.3f | %.3f | %.3f |" % (
            r["launch"], r["layer"], r["workgroups"], r["avg_us"], r["tflops"], r["frac_of_peak"],
            r["mfma_busy_frac"], r["wave_parked_frac"], r["issue_stall_frac"]))
    t = d["total"]
    out.append("")
    out.append("All 17 launches: %.2f ms per step, %.0f TFLOP/s = **%.3f** of the dense f16 peak on the box of the profile run "
               "(%d / %d steady-state steps averaged; `csrc` fingerprint `%s`)." % (
                   t["us_per_step"] / 1e3, t["tflops"], t["frac_of_peak"], d["steps_averaged"], d["pmc_steps_averaged"],
                   d["_provenance"]["csrc_sha16"]))
    return "\n".join(out)


def step_table(n, steps=25, top=16):
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", n + "_bench_kernel_stats.csv"))))
    agg = {}
    for r in rows:
        a = agg.setdefault(short(r["Name"]), [0.0, 0.0])
        a[0] += int(r["Calls"]) / steps
        a[1] += float(r["TotalDurationNs"]) / steps / 1e6
    total = sum(v[1] for v in agg.values())
    out = ["| kernel | launches / step | ms / step | share |", "|---|---|---|---|"]
    items = sorted(agg.items(), key=lambda kv: -kv[1][1])
    for k, v in items[:top]:
        out.append("| `%s` | %.0f | %.3f | %.3f |" % (k, v[0], v[1], v[1] / total))
    rest = items[top:]
    out.append("| %d more kernels | %.0f | %.3f | %.3f |" % (len(rest), sum(v[0] for _, v in rest), sum(v[1] for _, v in rest),
                                                          sum(v[1] for _, v in rest) / total))
    small = [(int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / steps / 1e6) for r in rows if float(r["AverageNs"]) < 60000]
    out.append("")
    out.append("%.2f ms of kernel time per step in all (steady state, %d steps); launches under 60 µs: %.0f per step = %.2f ms." % (
        total, steps, sum(s[0] for s in small), sum(s[1] for s in small)))
    tr = json.load(open(os.path.join(ROOT, "profiles", n + "_pmc_traffic.json")))
    w4 = tr["conv3x3_w4_kernel"]["launches_profiled"] / 17.0
    gb = sum(v["hbm_bytes_per_launch"] * v["launches_profiled"] / w4 / 1e9 for k, v in tr.items() if not k.startswith("_"))
    out.append("HBM traffic of the whole step (`profiles/%s_pmc_traffic.json` × launches): %.1f GB." % (n, gb))
    return "\n".join(out)


def main(n):
    p = os.path.join(ROOT, "DESIGN.md")
    s = open(p).read()
    for name, text in (("w4_per_layer", w4_table(n)), ("step_kernels", step_table(n))):
        pat = re.compile(r"(<!-- generated: %s -->\n).*?(\n<!-- end generated -->)" % name, re.S)
        if not pat.search(s):
            raise SystemExit("marker for %s not found in DESIGN.md" % name)
        s = pat.sub(lambda m: m.group(1) + text + m.group(2), s)
    open(p, "w").write(s)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r03")
