"""One training step out of a rocprofv3 --kernel-trace CSV: wall time between two
conv_pack_many launches, busy time (union over queues), per-queue busy, idle
gaps, the longest gaps and what follows them, kernel time by family.
  python tools/step_timeline.py kernel_trace.csv [step_index_from_end=1]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
K = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in rows]
K.sort()
marks = [i for i, k in enumerate(K) if 'conv_pack_many' in k[2]]
a, b = marks[-1 - back], marks[-back]
S = K[a + 1:b + 1]
t0, t1 = K[a][1], K[b][1]
print('step: %d kernels, wall %.3f ms' % (len(S), (t1 - t0) / 1e6))
# union busy
ev = sorted((s, e) for s, e, _, _ in S)
busy = 0; cur_s, cur_e = ev[0]
gaps = []
for s, e in ev[1:]:
  if s > cur_e:
    busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e)); cur_s, cur_e = s, e
  else:
    cur_e = max(cur_e, e)
busy += cur_e - cur_s
print('GPU busy (union) %.3f ms, idle %.3f ms in %d gaps; sum of kernel durations %.3f ms'
      % (busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps), sum(e - s for s, e, _, _ in S) / 1e6))
q = collections.Counter()
for s, e, n, qq in S: q[qq] += e - s
print('kernel time per queue (ms):', {k: round(v / 1e6, 3) for k, v in q.items()})
hist = collections.Counter()
for g, _ in gaps:
  hist[min(int(g / 1000) , 50)] += g
print('idle by gap length (us bucket: total ms):', {k: round(v / 1e6, 3) for k, v in sorted(hist.items())})
fam = collections.Counter(); cnt = collections.Counter()
def family(n):
  for key in ('bn_', 'conv_wgrad', 'conv3x3_wgrad', 'conv_igemm', 'conv_splitk', 'conv3x3_c32', 'pred_', 'conv_first', 'conv_pack',
              'splat', 'disp_reg', 'zbuf', 'view_synth', 'multi_tensor', 'elementwise', 'copy', 'Cat'):
    if key in n: return key
  return 'other'
for s, e, n, _ in S:
  fam[family(n)] += e - s; cnt[family(n)] += 1
for k, v in fam.most_common():
  print('  %-16s %4d launches %8.3f ms  avg %6.1f us' % (k, cnt[k], v / 1e6, v / 1e3 / cnt[k]))
if len(sys.argv) > 3:
  for s, e, n, qq in S:
    print('%9.1f %7.1f q%s %s' % ((s - t0) / 1e3, (e - s) / 1e3, qq, n[:90]))
