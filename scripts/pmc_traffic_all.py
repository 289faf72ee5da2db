"""Per-kernel HBM traffic of one bench.py run from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), joined
with the average launch durations of a --kernel-trace --stats run of the same command:

    python scripts/pmc_traffic_all.py <fetch_dir> <write_dir> <kernel_stats.csv> > profiles/r01_pmc_traffic_all.json

Counter units and the gfx950 FETCH_SIZE x2 correction as in scripts/pmc_traffic.py (MI355X_MICROARCH.md, HBM)."""
import collections, csv, glob, json, re, sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:90]


acc = {}
for counter, d in (("FETCH_SIZE", sys.argv[1]), ("WRITE_SIZE", sys.argv[2])):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            per[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    acc[counter] = per
dur = {}
for r in csv.DictReader(open(sys.argv[3])):
    dur[short(r["Name"])] = (float(r["AverageNs"]), int(r["Calls"]), float(r["Percentage"]))
res = {}
for k in sorted(set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"]), key=lambda k: -dur.get(k, (0, 0, 0))[2]):
    fv, wv = acc["FETCH_SIZE"].get(k, []), acc["WRITE_SIZE"].get(k, [])
    rd = (sum(fv) / len(fv) if fv else 0.0) * 1024 * 2
    wr = (sum(wv) / len(wv) if wv else 0.0) * 1024
    e = {"launches_profiled": len(fv), "hbm_read_bytes_per_launch": round(rd), "hbm_write_bytes_per_launch": round(wr)}
    if k in dur and dur[k][0] > 0:
        e["avg_launch_us"] = round(dur[k][0] / 1e3, 1)
        e["share_of_kernel_time_pct"] = dur[k][2]
        e["hbm_GBps_at_avg_duration"] = round((rd + wr) / dur[k][0], 1)
    res[k] = e
print(json.dumps(res, indent=1))
