"""Platform-stable synthetic images for the full-size parity fixtures.

The large goldens (512^2, 1024^2) cannot store their input images (3 x 12 MB each), so the images are
regenerated from a seed wherever the test runs.  torch's bicubic resize and `randn` are not bit-stable
across host CPUs (vectorised code paths differ), so this generator uses integer hashing (the same
splitmix64 as the synthetic VGG weights) and exactly-rounded float64 numpy arithmetic only: every
machine produces the same bits.  Each fixture stores float64 checksums of its inputs to prove it.
"""
import numpy as np
import torch

_MASK64 = (1 << 64) - 1


def _splitmix64(x):
    x = x + np.uint64(0x9E3779B97F4A7C15)
    z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def hash_uniform(count, stream):
    """`count` float64 uniforms in [0, 1) from a counter-based hash (no RNG state, no libm)."""
    idx = np.arange(count, dtype=np.uint64)
    with np.errstate(over='ignore'):
        base = np.uint64(((stream + 1) * 0xD6E8FEB86659FD93) & _MASK64)
        z = _splitmix64(idx * np.uint64(0x2545F4914F6CDD1D) + base)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def smooth_image(seed, h, w):
    """Photo-like field in [0, 1] as a [1, 3, h, w] fp32 tensor: uniform noise on a 1/16-resolution grid,
    bilinearly upsampled, plus +-12/255 of fine noise (the recipe of SURVEY.md 8(d), with a bilinear
    instead of a bicubic upsample so that it is exact IEEE arithmetic)."""
    lh, lw = h // 16 + 2, w // 16 + 2
    low = hash_uniform(3 * lh * lw, seed * 2).reshape(3, lh, lw)
    ys = np.arange(h, dtype=np.float64) * ((lh - 1) / max(h - 1, 1))
    xs = np.arange(w, dtype=np.float64) * ((lw - 1) / max(w - 1, 1))
    y0 = np.minimum(np.floor(ys).astype(np.int64), lh - 2)
    x0 = np.minimum(np.floor(xs).astype(np.int64), lw - 2)
    fy = (ys - y0)[None, :, None]
    fx = (xs - x0)[None, None, :]
    a = low[:, y0][:, :, x0]
    b = low[:, y0][:, :, x0 + 1]
    c = low[:, y0 + 1][:, :, x0]
    d = low[:, y0 + 1][:, :, x0 + 1]
    top = a + (b - a) * fx
    bot = c + (d - c) * fx
    img = top + (bot - top) * fy
    noise = (hash_uniform(3 * h * w, seed * 2 + 1).reshape(3, h, w) - 0.5) * (24.0 / 255.0)
    img = np.clip(img + noise, 0.0, 1.0).astype(np.float32)
    return torch.from_numpy(img)[None].contiguous()


def checksum(t):
    """Exact, order-independent fingerprint of a fp32 tensor - stored in the fixtures to detect generator drift:
    integer sums of the raw bit patterns (a float64 sum depends on the reduction order, i.e. on the host's thread
    count and vector width: it differed in the last bit between two GPU boxes)."""
    bits = t.contiguous().view(torch.int32).to(torch.int64).flatten()
    idx = torch.arange(bits.numel(), dtype=torch.int64) % 127 + 1        # no int64 overflow below 6e7 elements
    return np.array([int(bits.sum()), int((bits * idx).sum() % (1 << 61))], dtype=np.int64)
