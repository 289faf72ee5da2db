for r in 1 2; do
for c in 1.5 1.8 2.1 2.5 3.0; do
  OCR_GUEST_COVER=$c python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-config-legs --no-proxy 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cover $c r$r: %.3f ms' % d['ms_per_step'])"
done
done
for r in 1 2; do
for m in 20 40 70 120; do
  OCR_GUEST_MIN_US=$m python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-config-legs --no-proxy 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('min_us $m r$r: %.3f ms' % d['ms_per_step'])"
done
done
