#!/usr/bin/env python3
"""Per-kernel means of the PMC passes of tools/pmc_ns.sh over the n = 512 Newton-Schulz launches ->
profiles/rNN_pmc_ns512.md.      python tools/pmc_ns_summary.py gpurun_out/pmcns"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
KERNELS = ('gemm_staged_kernel', 'ns_gemm_f16_kernel')
table = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ('a', 'b', 'c'):
    for path in glob.glob(os.path.join(root, sub, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                name = next((k for k in KERNELS if k in r['Kernel_Name']), None)
                if name:
                    table[name][r['Counter_Name']].append(float(r['Counter_Value']))
print('| counter (mean per launch) | ' + ' | '.join(f'`{k}<512, 4>`' for k in KERNELS) + ' |')
print('|---|' + '---:|' * len(KERNELS))
counters = sorted({c for k in KERNELS for c in table[k]})
for c in counters:
    cells = []
    for k in KERNELS:
        v = table[k].get(c)
        cells.append(f'{sum(v) / len(v):,.0f} ({len(v)})' if v else '-')
    print(f'| {c} | ' + ' | '.join(cells) + ' |')
for k in KERNELS:
    t = table[k]
    mean = lambda c: sum(t[c]) / len(t[c]) if t.get(c) else None
    hit, miss = mean('TCC_HIT_sum'), mean('TCC_MISS_sum')
    if hit is not None and miss is not None and hit + miss > 0:
        print(f'\n`{k}`: L2 hit rate {hit / (hit + miss):.1%}', end='')
    wave, mfma = mean('SQ_WAVE_CYCLES'), mean('SQ_VALU_MFMA_BUSY_CYCLES')
    if wave and mfma is not None:       # SQ_WAVE_CYCLES counts quad-cycles (4 clocks), the MFMA counter clocks; one wave per SIMD
        print(f'; matrix pipe busy {mfma / (4 * wave):.1%} of the waves\' resident time', end='')
    wave, wait = mean('SQ_WAVE_CYCLES'), mean('SQ_WAIT_INST_ANY')
    if wave and wait is not None:
        print(f'; waves waiting {wait / wave:.1%} of their cycles', end='')
    lds, conf = mean('SQ_ACTIVE_INST_LDS'), mean('SQ_LDS_BANK_CONFLICT')
    if lds and conf is not None:
        print(f'; LDS bank-conflict cycles {conf / lds:.1%} of LDS-active cycles', end='')
print()
