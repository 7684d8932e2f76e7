"""Per-kernel durations and the gaps between consecutive dispatches of one queue, from a rocprofv3 --kernel-trace CSV.
usage: python tools/trace_gaps.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def short(n):
    for k in ('k_iter_select2', 'k_iter_select', 'k_iter_update'):
        if k in n: return k
    return None
byq = collections.defaultdict(list)
for r in rows:
    k = short(r['Kernel_Name'])
    if k: byq[r['Queue_Id']].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), k))
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for q, l in byq.items():
    l.sort()
    for i, (s, e, k) in enumerate(l):
        dur[k].append((e - s) / 1e3)
        if i: gap[l[i - 1][2] + ' -> ' + k].append((s - l[i - 1][1]) / 1e3)
def stat(v):
    v = sorted(v); n = len(v)
    return 'n %6d  mean %6.2f  p50 %6.2f  p90 %6.2f  p99 %6.2f us' % (n, sum(v) / n, v[n // 2], v[int(n * .9)], v[int(n * .99)])
print('queues with loop kernels:', len(byq))
for k, v in dur.items(): print('duration %-16s %s' % (k, stat(v)))
for k, v in gap.items(): print('gap %-34s %s' % (k, stat(v)))
# along the chain: mean duration per 2000 dispatches of the first queue
q0 = sorted(byq.values(), key=len)[-1]
for k in dur:
    l = [(e - s) / 1e3 for s, e, kk in q0 if kk == k]
    print('along the chain (one queue, per 2000 launches)', k, ' '.join('%.1f' % (sum(l[i:i + 2000]) / max(len(l[i:i + 2000]), 1)) for i in range(0, len(l), 2000)))
