#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE of the 3x3 trunk conv launches from the two passes of tools/pmc_traffic.sh ->
profiles/rNN_pmc_traffic_conv.json (gfx950 correction: FETCH_SIZE x2, see MI355X_MICROARCH.md "HBM").
    python tools/pmc_traffic_summary.py gpurun_out/pmc_traffic profiles/r02_pmc_traffic_conv.json"""
import csv, glob, json, os, sys

root, out = sys.argv[1], sys.argv[2]
KERNELS = ('conv_split_kernel', 'conv_pc_kernel', 'conv_fat_kernel')


def per_launch(sub, counter):
    path = glob.glob(os.path.join(root, sub, '**', '*counter_collection.csv'), recursive=True)[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter and any(k in r['Kernel_Name'] for k in KERNELS):
            tot += float(r['Counter_Value'])
            n += 1
    return tot / max(n, 1), n


f, nf = per_launch('f', 'FETCH_SIZE')
w, nw = per_launch('w', 'WRITE_SIZE')
res = {
    'config': 'bench.py --size %s fp16x3, conv_split_kernel + conv_pc_kernel + conv_fat_kernel launches (3x3 trunk fwd + dgrad)' % os.environ.get('SIZE', '512'),
    'launches_sampled': nf,
    'FETCH_SIZE_KB_per_launch_raw': f, 'WRITE_SIZE_KB_per_launch_raw': w,
    'correction': 'FETCH_SIZE x2 for wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM)',
    'hbm_side_bytes_per_launch': (2 * f + w) * 1024,
    'note': 'memory-side (fabric) requests incl. Infinity-Cache hits; each of the 8 XCD L2s fetches the layer weights itself',
}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res))
