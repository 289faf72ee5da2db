#!/bin/bash
# Turns the passes of scripts/collect_profiles.sh (merged back under gpurun_out/) into the committed summaries.
#   bash scripts/make_profiles.sh gpurun_out/rNN rNN
set -e
R=${1:?collected directory}; N=${2:?round prefix, e.g. r03}
cd "$(dirname "$0")/.."
python scripts/steady_stats.py $R/stats > profiles/${N}_bench_kernel_stats.csv
cp $R/stats/*/*_kernel_stats.csv profiles/${N}_bench_kernel_stats_raw.csv
python scripts/pmc_mfma.py $R/pmc_m > profiles/${N}_pmc_mfma.json
python scripts/pmc_traffic.py $R/pmc_f $R/pmc_w > profiles/${N}_pmc_traffic.json
cp $R/clock_diag.json profiles/${N}_clock_diag.json
python scripts/w4_layers.py $R/stats $R/pmc_m > profiles/${N}_w4_per_layer.json
python - "$R" "$N" <<'P'
import json, sys
sys.path.insert(0, "scripts")
from pmc_mfma import provenance
R, N = sys.argv[1], sys.argv[2]
d = json.load(open(R + "/step_calls.json"))
out = {"_provenance": provenance("python3 scripts/step_calls.py (HIP events around every recorded call, median of 7 replays)")}
out.update(d)
json.dump(out, open("profiles/%s_step_calls.json" % N, "w"), indent=1)
P
python scripts/steady_stats.py $R/stats_resnet > profiles/${N}_resnet50_east_640_b64_kernel_stats.csv
OCR_STORAGE=bf16 python scripts/pmc_traffic.py $R/pmc_f_resnet $R/pmc_w_resnet "OCR_STORAGE=bf16 python3 scripts/bench_configs.py --which resnet --steps 3 --warmup 1 (3 engine-build + 1 warm-up + 3 timed + 1 loss step = 8 steps in the pass)" > profiles/${N}_resnet_pmc_traffic.json
python scripts/steady_stats.py $R/stats_pl --marker momentum_kernel > profiles/${N}_pixellink_vgg_512_b32_kernel_stats.csv
cp $R/stats_dec/*/*_kernel_stats.csv profiles/${N}_decode_lanms_kernel_stats.csv
python - <<'P'
import json
from tensorflow_ocr_amd import _lib
now = _lib.csrc_fingerprint()
for f in ("pmc_mfma", "pmc_traffic", "clock_diag", "w4_per_layer", "step_calls"):
    import glob
    for p in sorted(glob.glob("profiles/*_%s.json" % f))[-1:]:
        print(p, json.load(open(p))["_provenance"]["csrc_sha16"], "current" if json.load(open(p))["_provenance"]["csrc_sha16"] == now else "STALE (sources changed since)")
P
# round 5: co-residency probe, real-kernel guest probe, the recorded step's guest / host pairs, serial-vs-guests trace diff
for f in coresidency_probe.json guest_probe.json guest_pairs.json guest_pairs_trace.txt trace_diff_serial_vs_guests.txt; do
  [ -f $R/$f ] && cp $R/$f profiles/${N}_$f
done
python scripts/design_tables.py $N
