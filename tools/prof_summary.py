#!/usr/bin/env python3
"""Summarise a rocprofv3 run (rocpd sqlite database or *_kernel_stats.csv) as a markdown table.

    python tools/prof_summary.py gpurun_out/prof/bench_results.db [steps] > profiles/rNN_bench_kernels.md
"""
import csv
import sqlite3
import sys


def rows_from_db(path):
    db = sqlite3.connect(path)
    return [(r[0], int(r[1]), float(r[2]), float(r[3]), float(r[4]))
            for r in db.execute('select name, total_calls, total_duration, average, percentage from top_kernels')]


def rows_from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r['Name'], int(r['Calls']), float(r['TotalDurationNs']) / 1e3,
                        float(r['AverageNs']) / 1e3, float(r['Percentage'])))
    return out


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
    rows = rows_from_db(path) if path.endswith('.db') else rows_from_csv(path)
    rows.sort(key=lambda r: -r[2])
    total = sum(r[2] for r in rows)
    print(f'source: {path}   total kernel time {total / 1e3:.3f} ms' +
          (f'   ({total / 1e3 / steps:.3f} ms per step over {steps} steps)' if steps else ''))
    print()
    print('| kernel | calls | total us | avg us | % |')
    print('|---|---:|---:|---:|---:|')
    for name, calls, tot, avg, pct in rows:
        name = name.replace('st::(anonymous namespace)::', '').replace('|', '\\|')
        if len(name) > 110:
            name = name[:107] + '...'
        print(f'| `{name}` | {calls} | {tot:.1f} | {avg:.2f} | {pct:.2f} |')


if __name__ == '__main__':
    main()
