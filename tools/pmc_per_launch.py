#!/usr/bin/env python3
"""Per-launch FETCH_SIZE (x2, gfx950 correction) / WRITE_SIZE of the convolution launches of one bench step, from the two
passes of tools/pmc_traffic.sh:  python tools/pmc_per_launch.py gpurun_out/pmc_traffic > table.md"""
import csv, glob, os, sys
root = sys.argv[1]


def rows(sub, counter):
    path = glob.glob(os.path.join(root, sub, '**', '*counter_collection.csv'), recursive=True)[0]
    out = []
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter and any(k in r['Kernel_Name'] for k in ('conv_pc_kernel', 'conv_split_kernel', 'conv_fat_kernel')):
            out.append((int(r['Dispatch_Id']), r['Kernel_Name'].split('::')[-1][:34], int(r['Grid_Size']), float(r['Counter_Value'])))
    return sorted(out)


f, w = rows('f', 'FETCH_SIZE'), rows('w', 'WRITE_SIZE')
n = min(len(f), len(w))
tot = 0.0
for (i, name, grid, fv), (_, _, _, wv) in zip(f[-n:], w[-n:]):
    mb = (2 * fv + wv) / 1024
    tot += mb
    print(f'{i:6d} {name:34s} grid {grid:7d} fetch {2 * fv / 1024:7.1f} MB write {wv / 1024:6.1f} MB')
print(f'launches {n}, mean {tot / n:.2f} MB')
