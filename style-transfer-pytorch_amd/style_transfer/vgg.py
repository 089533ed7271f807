"""VGG-19 trunk description and weight handling (no torchvision dependency).

The hot path evaluates the first 30 entries of torchvision's ``vgg19().features`` (reference
``style_transfer.py:35``: ``models.vgg19(...).features[:layers[-1] + 1]`` with the default taps
``[1, 6, 11, 20, 22, 29]``, ``style_transfer.py:316-317``).  This module only *describes* that
network (layer indices, channel counts, pooling levels) and moves weights around; the arithmetic
lives in the HIP library (``csrc/``).
"""

import math

import torch

# (features index, kind, cin, cout).  Index numbering follows torchvision's cfg "E".
#   kind: 'conv' (3x3, stride 1, pad 1, followed by an in-place ReLU at index+1) or 'pool' (2x2/2).
LAYERS = [
    (0, 'conv', 3, 64), (2, 'conv', 64, 64), (4, 'pool', 64, 64),
    (5, 'conv', 64, 128), (7, 'conv', 128, 128), (9, 'pool', 128, 128),
    (10, 'conv', 128, 256), (12, 'conv', 256, 256), (14, 'conv', 256, 256), (16, 'conv', 256, 256),
    (18, 'pool', 256, 256),
    (19, 'conv', 256, 512), (21, 'conv', 512, 512), (23, 'conv', 512, 512), (25, 'conv', 512, 512),
    (27, 'pool', 512, 512),
    (28, 'conv', 512, 512),
]
CONV_INDICES = [i for i, kind, _, _ in LAYERS if kind == 'conv']          # 13 entries
CONV_SHAPES = [(cout, cin) for _, kind, cin, cout in LAYERS if kind == 'conv']
NUM_FEATURE_LAYERS = 30

# ImageNet statistics applied *after* the 'input' tap (reference style_transfer.py:30-31,84-85).
NORM_MEAN = (0.485, 0.456, 0.406)
NORM_STD = (0.229, 0.224, 0.225)

POOLINGS = ('max', 'average', 'l2')
POOLING_SCALES = {'max': 1., 'average': 2., 'l2': 0.78}   # reference style_transfer.py:22


def min_size_for(layers):
    """Smallest legal input edge for a set of taps (reference ``_get_min_size``, :61-69)."""
    last = max(layers)
    size = 1
    for pool_index_plus in (4, 9, 18, 27, 36):
        if last < pool_index_plus:
            break
        size *= 2
    return size


_MASK64 = (1 << 64) - 1


def _splitmix64(x):
    """Counter-based 64-bit mixer (splitmix64 finaliser) on a numpy uint64 array; wraps mod 2^64."""
    import numpy as np
    x = x + np.uint64(0x9E3779B97F4A7C15)
    z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _approx_normal(count, stream):
    """Deterministic ~N(0,1) samples: sqrt(3) * (sum of 4 uniforms - 2), from integer hashing only.

    torch.randn / numpy's ziggurat are NOT bit-stable across host CPUs (vectorised code paths), which
    would silently invalidate the golden fixtures on another machine.  Everything here is integer
    arithmetic plus exactly-rounded IEEE adds/multiplies, so every platform produces the same bits."""
    import numpy as np
    idx = np.arange(count, dtype=np.uint64)
    acc = np.zeros(count, dtype=np.float64)
    with np.errstate(over='ignore'):
        for k in range(4):
            base = np.uint64(((stream * 4 + k + 1) * 0xD6E8FEB86659FD93) & _MASK64)
            z = _splitmix64(idx * np.uint64(0x2545F4914F6CDD1D) + base)
            acc += (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return (acc - 2.0) * 1.7320508075688772


def synthetic_vgg19_weights(seed=0, dtype=torch.float32):
    """Seeded stand-in for ``vgg19-dcbb9e9d.pth`` (no network, no pretrained file in this image).

    He-scaled conv weights and small biases from a platform-independent generator so that the
    oracle, the golden fixtures and the HIP path see bit-identical parameters on every machine.
    Returns a list of 13 ``(weight[cout, cin, 3, 3], bias[cout])`` CPU tensors.
    """
    import numpy as np
    params = []
    for layer, (cout, cin) in enumerate(CONV_SHAPES):
        std = math.sqrt(2.0 / (cin * 9))
        w = _approx_normal(cout * cin * 9, (seed * 64 + layer) * 2) * std
        b = _approx_normal(cout, (seed * 64 + layer) * 2 + 1) * 0.05
        w = torch.from_numpy(w.astype(np.float32)).reshape(cout, cin, 3, 3)
        b = torch.from_numpy(b.astype(np.float32))
        params.append((w.to(dtype), b.to(dtype)))
    return params


def weights_from_state_dict(state_dict):
    """Pick the 13 trunk convs out of a torchvision ``vgg19`` state dict.

    Accepts both the full-model keys (``features.0.weight``) and bare ``vgg19().features`` keys
    (``0.weight``).  This is the loader a user points at their own ``vgg19-dcbb9e9d.pth``.
    """
    params = []
    for idx in CONV_INDICES:
        for prefix in (f'features.{idx}.', f'{idx}.'):
            if prefix + 'weight' in state_dict:
                w = state_dict[prefix + 'weight'].detach().to(torch.float32).cpu().contiguous()
                b = state_dict[prefix + 'bias'].detach().to(torch.float32).cpu().contiguous()
                break
        else:
            raise KeyError(f'VGG-19 state dict has no conv parameters for features[{idx}]')
        params.append((w, b))
    for (w, b), (cout, cin) in zip(params, CONV_SHAPES):
        if tuple(w.shape) != (cout, cin, 3, 3) or tuple(b.shape) != (cout,):
            raise ValueError('state dict does not hold VGG-19 (cfg E) convolution shapes')
    return params


def load_weights(path=None, seed=0):
    """``path`` -> real torchvision checkpoint; ``None`` -> seeded synthetic weights."""
    if path is None:
        return synthetic_vgg19_weights(seed)
    return weights_from_state_dict(torch.load(path, map_location='cpu', weights_only=True))


def weights_fingerprint(params):
    """Cheap content hash used by the golden fixtures to detect RNG drift."""
    acc = []
    for w, b in params:
        acc.append(float(w.double().sum()))
        acc.append(float(w.double().abs().sum()))
        acc.append(float(b.double().sum()))
    return acc
