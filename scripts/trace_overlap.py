"""Dev: from a rocprofv3 --kernel-trace CSV, measure per training step: wall, GPU-idle time, time with
>=2 kernels resident, and the busy time per stream (queue)."""
import csv, glob, sys, re, collections
d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', r.get('Stream_Id', '0'))))
ev.sort()
# steps are delimited by adam_kernel launches
adam = [i for i, e in enumerate(ev) if 'adam_kernel' in e[2]]
print('kernels', len(ev), 'adam launches', len(adam))
if len(adam) < 3:
    sys.exit()
lo, hi = adam[-3] + 1, adam[-1] + 1       # the last two full steps
seg = ev[lo:hi]
t0, t1 = seg[0][0], max(e[1] for e in seg)
pts = []
for s, e, n, q in seg:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
busy = over = 0; cur = 0; last = t0
for t, dlt in pts:
    if cur >= 1: busy += t - last
    if cur >= 2: over += t - last
    cur += dlt; last = t
wall = (t1 - t0) / 2e6
print('per step: wall %.2f ms, busy %.2f, idle %.2f, >=2 kernels %.2f' % (wall, busy / 2e6, wall - busy / 2e6, over / 2e6))
perq = collections.defaultdict(float)
for s, e, n, q in seg: perq[q] += (e - s) / 2e6
print('busy per queue (ms/step):', {k: round(v, 2) for k, v in perq.items()})
# biggest idle gaps
gaps = []
cur = 0; last = t0
for t, dlt in pts:
    if cur == 0 and t - last > 20000: gaps.append((t - last, last))
    cur += dlt; last = t
gaps.sort(reverse=True)
for g, at in gaps[:12]:
    prev = max((e for e in seg if e[1] <= at + 1), key=lambda e: e[1], default=None)
    nxt = min((e for e in seg if e[0] >= at + g - 1), key=lambda e: e[0], default=None)
    print('gap %.1f us after %s before %s' % (g / 1e3, re.sub(r'.*N_1\d+', '', prev[2])[:40] if prev else None, re.sub(r'.*N_1\d+', '', nxt[2])[:40] if nxt else None))
# per-queue totals by kernel (ms/step)
byk = collections.defaultdict(float)
for s, e, n, q in seg:
    n2 = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', n); n2 = re.sub(r'\(anonymous namespace\)::', '', n2)
    n2 = re.sub(r'EvNS_.*|\(.*', '', n2)
    byk[(q, n2[:48])] += (e - s) / 2e6
for (q, n), v in sorted(byk.items(), key=lambda kv: -kv[1])[:28]:
    print('q%s %-48s %.3f' % (q, n, v))
