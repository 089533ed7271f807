#!/usr/bin/env python3
"""Tile-shape sweep of the producer / consumer convolution (st_conv_pc.hip): every trunk layer at the given image
sizes, forward and data gradient, under each forced tile (ST_CONV_PC_SHAPE / _TW / _KSPLIT), the round-1 selection
rule (ST_CONV_PC_MODEL=0), the cost model (default) and the single-role kernel (ST_CONV_PC=0).
    python tools/conv_shapes.py 128 181 256 362 512 > profiles/...      (one line per layer and direction)"""
import ctypes
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'style-transfer-pytorch_amd'))
import torch                                   # noqa: E402
from style_transfer import _hip                # noqa: E402

lib = _hip.load_library()
LAYERS = [('conv1_2', 64, 64, 0, 1), ('conv2_1', 64, 128, 1, 1), ('conv2_2', 128, 128, 1, 1), ('conv3_1', 128, 256, 2, 1),
          ('conv3_2', 256, 256, 2, 3), ('conv4_1', 256, 512, 3, 1), ('conv4_2', 512, 512, 3, 3), ('conv5_1', 512, 512, 4, 1)]
ITERS = 20


def timed(cin, cout, h, w, dgrad):
    us = ctypes.c_double()
    rc = lib.st_op_conv3x3_time(cin, cout, h, w, dgrad, 4, ITERS, ctypes.byref(us), None)
    return us.value if rc == 0 else float('nan')


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [362]
    forced = [(1, 32, 1)] + [(s, tw, ks) for s in (2, 3) for tw in (32, 16, 8) for ks in (1, 2, 4, 8, 16)]
    for size in sizes:
        totals = {'rule': 0.0, 'model': 0.0, 'best': 0.0, 'single': 0.0, 'default': 0.0, 'nosplit': 0.0}
        for name, cin, cout, lvl, mult in LAYERS:
            h = size >> lvl
            for dgrad in (0, 1):
                if dgrad and name == 'conv5_1':
                    continue                                   # masked in the plan: not a producer/consumer layer
                ci, co = (cout, cin) if dgrad else (cin, cout)
                res = {}
                for shape, tw, ks in forced:
                    if ks > 1 and ((ci // 16) % ks or ci // 16 // ks < 2):
                        continue
                    with _hip.options(ST_CONV_PC=2, ST_CONV_PC_SHAPE=shape, ST_CONV_PC_TW=tw, ST_CONV_PC_KSPLIT=ks,
                                      ST_CONV_NOMASK=1):
                        res[(shape, tw, ks)] = timed(cin, cout, h, h, dgrad)
                with _hip.options(ST_CONV_PC_MODEL=0, ST_CONV_NOMASK=1):
                    rule = timed(cin, cout, h, h, dgrad)
                with _hip.options(ST_CONV_PC=2, ST_CONV_NOMASK=1):
                    model = timed(cin, cout, h, h, dgrad)
                with _hip.options(ST_CONV_PC=0, ST_CONV_NOMASK=1):
                    single = timed(cin, cout, h, h, dgrad)
                with _hip.options(ST_CONV_NOMASK=1):
                    default = timed(cin, cout, h, h, dgrad)
                with _hip.options(ST_CONV_NOMASK=1, ST_CONV_PC_SPLIT=0):
                    nosplit = timed(cin, cout, h, h, dgrad)
                best = min(res, key=lambda k: res[k])
                totals['rule'] += mult * rule
                totals['model'] += mult * min(model, default)
                totals['best'] += mult * min(res[best], single)
                totals['single'] += mult * single
                totals['default'] += mult * default
                totals['nosplit'] += mult * nosplit
                cells = ' '.join(f'{k[0]}/{k[1]}/{k[2]}={v:.1f}' for k, v in sorted(res.items()))
                print(f'{size} {name} {"dgrad" if dgrad else "fwd"} {ci}->{co} @{h}: rule {rule:.1f} model(pc) {model:.1f} '
                      f'default {default:.1f} nosplit {nosplit:.1f} single {single:.1f} best {best[0]}/{best[1]}/{best[2]}={res[best]:.1f} | {cells}',
                      flush=True)
        print(f'{size} TOTAL us (23 trunk convs): round-1 rule {totals["rule"]:.0f}, default now {totals["model"]:.0f}, '
              f'best forced {totals["best"]:.0f}, single-role kernel only {totals["single"]:.0f}; shipped default '
              f'{totals["default"]:.0f}, the same without two-shape covers {totals["nosplit"]:.0f}', flush=True)


if __name__ == '__main__':
    torch.cuda.set_device(0)
    main()
