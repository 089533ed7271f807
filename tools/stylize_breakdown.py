#!/usr/bin/env python3
"""Where the drop-in stylize() call of the default run (128 -> 512, 1000 + 500 x 4 iterations, Adam) spends its wall time:
per scale, setup (resample, plan, targets, range guard) against the iteration loop (ST_STYLIZE_TIMING=1).
    python tools/stylize_breakdown.py [end_scale]"""
import contextlib, io, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'style-transfer-pytorch_amd'))
os.environ['ST_STYLIZE_TIMING'] = '1'
import numpy as np
import torch
from PIL import Image
from style_transfer import StyleTransfer, vgg
end = int(sys.argv[1]) if len(sys.argv) > 1 else 512


def img(seed, size):
    rng = np.random.default_rng(seed)
    low = rng.random((size // 16, size // 16, 3))
    return Image.fromarray((np.kron(low, np.ones((16, 16, 1))) * 255).astype(np.uint8))


st = StyleTransfer(devices=['cuda:0'], pooling='max', weights=vgg.synthetic_vgg19_weights(0))
for attempt in ('first call', 'second call'):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        st.stylize(img(1, end), [img(2, end)], end_scale=end)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    setup = sum(t['setup_s'] for t in st.timing)
    loop = sum(t['loop_s'] for t in st.timing)
    print(f'[stylize_breakdown] {attempt}: {total:.3f} s = setup {setup:.3f} + loops {loop:.3f} + rest {total - setup - loop:.3f}')
    for t in st.timing:
        print(f"[stylize_breakdown]   {t['size'][0]}x{t['size'][1]}: setup {1e3 * t['setup_s']:.1f} ms, "
              f"{t['iterations']} iterations in {t['loop_s']:.3f} s = {t['iterations'] / t['loop_s']:.1f} it/s")
