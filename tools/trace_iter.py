#!/usr/bin/env python3
"""Per-stream timeline of the LAST full iteration in a rocprofv3 kernel trace of bench.py
(gpurun_out/trace/t_kernel_trace.csv): for each queue, its kernels with start/duration relative to the
iteration's first kernel (conv_first_fwd).   python tools/trace_iter.py [csv] [queue-to-list]"""
import csv, sys, collections, re
path = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/trace/t_kernel_trace.csv'
rows = [r for r in csv.DictReader(open(path))]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
starts = [i for i, r in enumerate(rows) if 'conv_first_fwd' in r['Kernel_Name']]
a, b = starts[-2], starts[-1]          # last complete iteration
it = rows[a:b]
t0 = it[0]['s']
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n); n = n.replace('st::', '')
    return n.split('(')[0][:60]
byq = collections.defaultdict(list)
for r in it:
    byq[r['Queue_Id']].append(r)
print(f'iteration span {(it[-1]["e"] - t0) / 1e3:.1f} us, {len(it)} kernels')
for q, rs in sorted(byq.items(), key=lambda kv: kv[1][0]['s']):
    busy = sum(r['e'] - r['s'] for r in rs) / 1e3
    print(f'queue {q}: {len(rs)} kernels, first start {(rs[0]["s"] - t0) / 1e3:8.1f} us, last end {(rs[-1]["e"] - t0) / 1e3:8.1f} us, busy {busy:7.1f} us')
want = sys.argv[2] if len(sys.argv) > 2 else None
if want:
    prev_end = None
    for r in byq[want]:
        gap = (r['s'] - prev_end) / 1e3 if prev_end else 0.0
        print(f'  +{(r["s"] - t0) / 1e3:8.1f} us  dur {(r["e"] - r["s"]) / 1e3:6.1f}  gap {gap:6.1f}  grid {r["Grid_Size_X"]:>7}x{r["Grid_Size_Y"]} wg {r["Workgroup_Size_X"]:>4}  {short(r["Kernel_Name"])}')
        prev_end = r['e']
