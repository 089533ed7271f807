#!/usr/bin/env python3
"""Timeline of one strip iteration over the real transport from a rocprofv3 kernel trace of tools/fabric_host_time.py:
per hardware queue its kernels, and the trunk queue's kernel list with gaps.  python tools/trace_strip.py trace.csv [iteration]"""
import csv, re, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
starts = [i for i, r in enumerate(rows) if 'conv_first_fwd' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 12
it = rows[starts[k]:starts[k + 1]] if k >= 0 else rows[starts[k - 1]:starts[k]]      # (negative: counted from the end)
t0 = it[0]['s']


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    return re.sub(r'^void ', '', n).replace('st::', '').split('(')[0][:44]


print('iteration span %.1f us, %d kernels' % ((it[-1]['e'] - t0) / 1e3, len(it)))
byq = collections.defaultdict(list)
for r in it:
    byq[r['Queue_Id']].append(r)
trunk = None
for q, rs in sorted(byq.items(), key=lambda kv: kv[1][0]['s']):
    names = collections.Counter(short(r['Kernel_Name']) for r in rs).most_common(3)
    print('queue %s: %3d kernels, busy %7.1f us, first %7.1f last %7.1f  %s' % (q, len(rs), sum(r['e'] - r['s'] for r in rs) / 1e3,
          (rs[0]['s'] - t0) / 1e3, (rs[-1]['e'] - t0) / 1e3, names))
    if any('conv_pc_kernel' in r['Kernel_Name'] for r in rs):
        trunk = q
prev = None
gaps = 0.0
for r in byq[trunk]:
    gap = (r['s'] - prev) / 1e3 if prev else 0.0
    if gap > 15:
        print('  trunk gap %6.1f us before +%8.1f %s' % (gap, (r['s'] - t0) / 1e3, short(r['Kernel_Name'])))
        gaps += gap
    prev = r['e']
print('trunk queue: gaps > 15 us sum to %.1f us' % gaps)
rc = [r for r in it if 'rccl' in r['Kernel_Name'].lower() or 'nccl' in r['Kernel_Name'].lower()]
print('%d RCCL kernels, %.1f us in total, mean %.1f us; on queues %s' % (len(rc), sum(r['e'] - r['s'] for r in rc) / 1e3,
      sum(r['e'] - r['s'] for r in rc) / 1e3 / max(len(rc), 1), sorted({r['Queue_Id'] for r in rc})))
