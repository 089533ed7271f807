#!/usr/bin/env python3
"""Wall time of the drop-in StyleTransfer.stylize() (host loop included) at one scale.
    python tools/stylize_time.py [size] [iterations]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'style-transfer-pytorch_amd'))
import torch
from PIL import Image
import numpy as np
from style_transfer import StyleTransfer, vgg
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(0)
def img(seed):
    rng = np.random.default_rng(seed)
    low = rng.random((size // 16, size // 16, 3))
    return Image.fromarray((np.kron(low, np.ones((16, 16, 1))) * 255).astype(np.uint8))
st = StyleTransfer(devices=['cuda:0'], pooling='max', weights=vgg.synthetic_vgg19_weights(0))
seen = []
def cb(it):
    seen.append(it.loss)
t0 = time.perf_counter()
st.stylize(img(1), [img(2)], min_scale=size, end_scale=size, iterations=iters, initial_iterations=iters, callback=cb)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'stylize {size}x{size}: {len(seen)} iterations in {dt:.3f} s -> {len(seen) / dt:.1f} it/s incl. setup; last loss {seen[-1]:.5f}')
t0 = time.perf_counter()
seen.clear()
st.stylize(img(1), [img(2)], min_scale=size, end_scale=size, iterations=iters, initial_iterations=iters, callback=cb)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'second call: {len(seen) / dt:.1f} it/s incl. target setup')
