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
rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats_dec -- python3 scripts/bench_configs.py --which decode > $R/stats_dec.log 2>&1 && echo decode ok
