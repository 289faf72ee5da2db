#!/bin/bash
# A/B of the headline step inside ONE gpurun call (box-to-box spread is +-2 %): alternating runs of bench.py's timed
# region under different environments.  usage: scripts/ab.sh "<env A>" "<env B>" [rounds] [steps]
A="$1"; B="$2"; R="${3:-2}"; S="${4:-30}"
mkdir -p gpurun_out
for r in $(seq 1 $R); do
  for which in A B; do
    if [ $which = A ]; then E="$A"; else E="$B"; fi
    env $E python bench.py --steps $S --warmup 5 --no-cpu-baseline --no-config-legs --no-proxy 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$which r$r [$E]: %.3f ms/step  %.1f img/s  w4 frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac']))"
  done
done
