"""Time the UNMODIFIED reference's `--devices cpu` path (staged by oracle/make_ref.py).  MEASUREMENT INFRASTRUCTURE:
imported only by bench.py's `cpu_baseline` leg; never by the product package.

Protocol = BASELINE.md section 3: `StyleTransfer(devices=['cpu'])`, the seeded synthetic VGG-19 weights copied into its
conv modules, `stylize(content, [style], min_scale = end_scale = S, initial_iterations = N, callback=...)`, the callback
collecting `STIterate.time` (the reference's own hook, style_transfer.py:492-493); it/s = 1 / median of the successive
differences after dropping the first two iterations.
"""
import contextlib
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, '_ref')


def available():
    return os.path.isfile(os.path.join(REF, 'style_transfer', 'style_transfer.py'))


class _Stop(Exception):
    pass


def _import_reference():
    """The staged reference under its own package name would collide with this repo's drop-in package
    (`style_transfer`), which bench.py has already imported: load it under the alias `ref_style_transfer`."""
    import importlib.util
    if 'ref_style_transfer' in sys.modules:
        return sys.modules['ref_style_transfer']
    shim = os.path.join(REF, 'shim')
    if shim not in sys.path:
        sys.path.insert(0, shim)                      # torchvision / tifffile stand-ins
    pkg_dir = os.path.join(REF, 'style_transfer')
    spec = importlib.util.spec_from_file_location('ref_style_transfer', os.path.join(pkg_dir, '__init__.py'),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules['ref_style_transfer'] = mod
    spec.loader.exec_module(mod)
    return mod


def _pil(t):
    from PIL import Image
    return Image.fromarray((t[0].permute(1, 2, 0) * 255).round().byte().numpy(), 'RGB')


def time_reference(size, weights, content, style, threads, max_iters=8, budget_s=8.0):
    """it/s of the reference at `threads` OpenMP threads; (its, n_timed, elapsed)."""
    ref = _import_reference()
    from ref_style_transfer import style_transfer as rst
    from style_transfer import vgg
    torch.set_num_threads(threads)
    stamps = []
    t_start = time.perf_counter()

    def callback(it):
        stamps.append(it.time)
        if len(stamps) >= max_iters or (len(stamps) >= 4 and time.perf_counter() - t_start > budget_s):
            raise _Stop()

    with contextlib.redirect_stdout(sys.stderr):       # the reference prints progress lines
        st = rst.StyleTransfer(devices=['cpu'])
        with torch.no_grad():
            for idx, (w, b) in zip(vgg.CONV_INDICES, weights):
                st.model.model[idx].weight.copy_(w)
                st.model.model[idx].bias.copy_(b)
        torch.manual_seed(0)
        try:
            st.stylize(_pil(content), [_pil(style)], min_scale=size, end_scale=size, initial_iterations=max_iters,
                       callback=callback)
        except _Stop:
            pass
    d = np.diff(np.array(stamps))[1:]                  # iterations 1 and 2 dropped
    if len(d) == 0:
        return 0.0, 0, time.perf_counter() - t_start
    return 1.0 / float(np.median(d)), len(d), time.perf_counter() - t_start
