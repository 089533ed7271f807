#!/usr/bin/env python3
"""Stage the UNMODIFIED reference for CPU-baseline timing on the GPU box.  TEST / MEASUREMENT INFRASTRUCTURE.

    python oracle/make_ref.py            # /root/reference/style_transfer -> oracle/_ref/style_transfer (verbatim)
                                         # tests/golden/_refshim          -> oracle/_ref/shim (torchvision / tifffile stand-ins)

`/root/reference` exists only in the build container; the GPU box receives a snapshot of this repo.  `oracle/_ref/` is
listed in .gitignore (reference sources never enter the history) but NOT in .gpurunignore, so the staged copy travels
with the snapshot exactly like the built libst_amd.so.  Only bench.py's `cpu_baseline` leg (through oracle/ref_runner.py)
imports it; the product package never does.  Nothing is modified: files are copied byte for byte and a manifest of
sha256 sums is written next to them (oracle/_ref/MANIFEST.json) so a reader can verify that.
"""
import hashlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SRC = '/root/reference/style_transfer'
SHIM = os.path.join(REPO, 'tests', 'golden', '_refshim')
DST = os.path.join(HERE, '_ref')


def _sha(path):
    h = hashlib.sha256()
    with open(path, 'rb') as f:
        h.update(f.read())
    return h.hexdigest()


def stage(verbose=True):
    if not os.path.isdir(SRC):
        if verbose:
            print(f'make_ref: {SRC} not present (GPU box?) - keeping whatever oracle/_ref already holds')
        return os.path.isdir(os.path.join(DST, 'style_transfer'))
    if os.path.isdir(DST):
        shutil.rmtree(DST)
    ignore = shutil.ignore_patterns('__pycache__', '*.pyc')
    shutil.copytree(SRC, os.path.join(DST, 'style_transfer'), ignore=ignore)
    shutil.copytree(SHIM, os.path.join(DST, 'shim'), ignore=ignore)
    manifest = {}
    for root, _, files in os.walk(os.path.join(DST, 'style_transfer')):
        for name in files:
            p = os.path.join(root, name)
            rel = os.path.relpath(p, os.path.join(DST, 'style_transfer'))
            manifest[rel] = {'sha256': _sha(p), 'same_as_reference': _sha(p) == _sha(os.path.join(SRC, rel))}
            os.chmod(p, 0o644)
    with open(os.path.join(DST, 'MANIFEST.json'), 'w') as f:
        json.dump({'source': SRC, 'files': manifest}, f, indent=1, sort_keys=True)
    assert all(v['same_as_reference'] for v in manifest.values())
    if verbose:
        print(f'make_ref: staged {len(manifest)} reference files (verbatim) + shim under {DST}')
    return True


if __name__ == '__main__':
    sys.exit(0 if stage() else 1)
