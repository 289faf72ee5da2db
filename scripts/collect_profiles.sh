#!/bin/bash
# Collects, ON THE GPU BOX, every trace / counter pass the files under profiles/ are made from.
#   gpurun --timeout 1150 -- 'bash scripts/collect_profiles.sh gpurun_out/rNN'
# then, back in the container:  bash scripts/make_profiles.sh gpurun_out/rNN rNN
# Counters in their own passes (never combined with sys / hip / hsa traces); the program itself after `--`.
set -o pipefail
R=${1:?output directory under gpurun_out/}
mkdir -p "$R"
export TMPDIR=/tmp
B="python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-config-legs --no-proxy"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config-legs --no-proxy > $R/stats.log 2>&1 && echo stats ok &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/pmc_f -- $B > $R/pmc_f.log 2>&1 && echo fetch ok &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/pmc_w -- $B > $R/pmc_w.log 2>&1 && echo write ok &&
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/pmc_m -- $B > $R/pmc_m.log 2>&1 && echo sq ok &&
python3 scripts/clock_diag.py > $R/clock_diag.json 2> $R/clock.log && echo clock ok &&
python3 scripts/step_calls.py > $R/step_calls.json 2> $R/step_calls.txt && echo calls ok &&
OCR_STORAGE=bf16 rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats_resnet -- python3 scripts/bench_configs.py --which resnet --steps 8 --warmup 3 > $R/stats_resnet.log 2>&1 && echo resnet ok &&
OCR_STORAGE=bf16 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/pmc_f_resnet -- python3 scripts/bench_configs.py --which resnet --steps 3 --warmup 1 > $R/pmc_f_resnet.log 2>&1 && echo resnet fetch ok &&
OCR_STORAGE=bf16 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/pmc_w_resnet -- python3 scripts/bench_configs.py --which resnet --steps 3 --warmup 1 > $R/pmc_w_resnet.log 2>&1 && echo resnet write ok &&
rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats_pl -- python3 scripts/bench_configs.py --which pixellink --steps 8 --warmup 3 > $R/stats_pl.log 2>&1 && echo pixellink ok &&
rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats_dec -- python3 scripts/bench_configs.py --which decode > $R/stats_dec.log 2>&1 && echo decode ok &&
# round 5: co-residency evidence and the guest / host pairs of the recorded step
mkdir -p scripts/_bin && hipcc --offload-arch=gfx950 -O3 -o scripts/_bin/coresidency_probe scripts/coresidency_probe.hip > $R/probe_build.log 2>&1 &&
hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o scripts/_bin/libguest.so scripts/guest_kernels.hip >> $R/probe_build.log 2>&1 &&
timeout -k 10 120 scripts/_bin/coresidency_probe > $R/coresidency_probe.json 2> $R/coresidency_probe.err && echo probe ok &&
timeout -k 10 300 python3 scripts/guest_probe.py > $R/guest_probe.log 2>&1 && cp gpurun_out/guest_probe.json $R/guest_probe.json && echo guest probe ok &&
rocprofv3 --kernel-trace --output-format csv -d $R/trace_guests -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-config-legs --no-proxy --no-pg > $R/trace_guests.log 2>&1 &&
OCR_GUEST_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d $R/trace_serial -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-config-legs --no-proxy --no-pg > $R/trace_serial.log 2>&1 &&
python3 scripts/guest_trace.py $R/trace_guests --dump > $R/guest_pairs_trace.txt && python3 scripts/guest_trace.py $R/trace_guests --json > $R/guest_pairs.json &&
python3 scripts/trace_diff.py $R/trace_serial $R/trace_guests > $R/trace_diff_serial_vs_guests.txt && rm -rf $R/trace_guests $R/trace_serial && echo pairs ok
