"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes for the conv_igemm kernels.
Per MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of the
bytes of a wide (16 B/lane) coalesced read stream -> doubled; WRITE_SIZE is exact for 16 B/lane stores."""
import csv, glob, json, re, sys, collections
out = {}
for name, d in (("FETCH_SIZE", sys.argv[1]), ("WRITE_SIZE", sys.argv[2])):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name and "conv_igemm" in r["Kernel_Name"]:
            k = re.search(r"conv_igemm_kernelILi(\d+)ELi(\d+)ELi(\d+)ELb([01])ELi(\d+)", r["Kernel_Name"])
            acc["conv_igemm_kernel<%s,%s,%s,%s,%s>" % k.groups()].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(k, {})[name] = {"launches": len(v), "avg_KiB": sum(v) / len(v)}
res = {}
for k, v in out.items():
    fetch = v.get("FETCH_SIZE", {}).get("avg_KiB", 0.0) * 1024 * 2      # gfx950 correction
    write = v.get("WRITE_SIZE", {}).get("avg_KiB", 0.0) * 1024
    res[k] = {"hbm_read_bytes_per_launch": fetch, "hbm_write_bytes_per_launch": write,
              "hbm_bytes_per_launch": fetch + write, "launches_profiled": v.get("FETCH_SIZE", {}).get("launches")}
print(json.dumps(res, indent=1))
