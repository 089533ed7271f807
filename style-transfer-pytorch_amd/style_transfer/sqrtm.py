"""Matrix square roots with backward passes - the module surface of reference ``style_transfer/sqrtm.py``.

``stylize()`` never comes through here: its Newton-Schulz chains run inside the plan (csrc/st_smallgemm.hip,
st_nsgemm.hip).  This module exists so that user code written against the reference
(``from style_transfer import sqrtm``; ``sqrtm.sqrtm_ns``, ``sqrtm.sqrtm_ns_lyap``, ``sqrtm.sqrtm_eig``) keeps
importing and gets the same numbers.  Tensors on a HIP device in the configuration the style loss uses (one
fp32 matrix of order 64 / 128 / 256 / 512, 12 iterations; reference style_transfer.py:154) go through the
library's standalone operators ``st_op_sqrtm_ns`` / ``st_op_sqrtm_ns_backward``; everything else (batches,
other iteration counts, float64, CPU tensors) is plain torch, step for step the recurrences of
reference sqrtm.py:9-25 and :36-47.
"""

import torch

_HIP_ORDERS = (64, 128, 256, 512)
_HIP_ITERS = 12


def _check_square(a):
    if a.ndim < 2:
        raise RuntimeError('tensor of matrices must have at least 2 dimensions')
    if a.shape[-2] != a.shape[-1]:
        raise RuntimeError('tensor must be batches of square matrices')


def _frobenius(a):
    return a.pow(2).sum(dim=[-2, -1], keepdim=True).sqrt()


def _hip_case(a, num_iters):
    return (a.is_cuda and a.dtype == torch.float32 and a.ndim == 2 and a.shape[-1] in _HIP_ORDERS
            and num_iters == _HIP_ITERS)


def sqrtm_ns(a, num_iters=10):
    """Newton-Schulz iteration for the principal square root (reference sqrtm.py:9-25): y0 = a / |a|_F, z0 = I,
    t = (3I - z y) / 2, y <- y t, z <- t z; the result is y sqrt(|a|_F).  Not differentiated specially - autograd
    unrolls it; use sqrtm_ns_lyap for the Lyapunov backward."""
    _check_square(a)
    if num_iters < 0:
        raise RuntimeError('num_iters must not be negative')
    if _hip_case(a, num_iters) and not (torch.is_grad_enabled() and a.requires_grad):
        from . import _hip
        return _hip.op_sqrtm_ns(a)
    norm = _frobenius(a)
    n = a.shape[-1]
    ident = torch.eye(n, device=a.device, dtype=a.dtype)
    three = ident * 3
    y = a / norm
    z = ident.repeat([*a.shape[:-2], 1, 1])
    for _ in range(num_iters):
        t = (three - z @ y) / 2
        y = y @ t
        z = t @ z
    return y * norm.sqrt()


class _MatrixSquareRootNSLyap(torch.autograd.Function):
    """Forward: sqrtm_ns.  Backward: the coupled iteration for the Lyapunov equation root X + X root = G
    (reference sqrtm.py:36-47), started from a = root / |root|_F, q = G / |root|_F."""

    @staticmethod
    def forward(ctx, a, num_iters, num_iters_backward):
        with torch.no_grad():
            root = sqrtm_ns(a, num_iters)
        ctx.save_for_backward(root)
        ctx.num_iters_backward = int(num_iters_backward)
        return root

    @staticmethod
    def backward(ctx, grad_output):
        root, = ctx.saved_tensors
        steps = ctx.num_iters_backward
        if _hip_case(root, steps) and grad_output.dtype == torch.float32:
            from . import _hip
            return _hip.op_sqrtm_ns_backward(root, grad_output.contiguous()), None, None
        norm = _frobenius(root)
        a = root / norm
        q = grad_output / norm
        three = torch.eye(root.shape[-1], device=root.device, dtype=root.dtype) * 3
        for i in range(steps):
            e = three - a @ a
            at = a.transpose(-2, -1)
            q = (q @ e - at @ (at @ q - q @ a)) / 2
            if i < steps - 1:
                a = a @ e / 2
        return q / 2, None, None


def sqrtm_ns_lyap(a, num_iters=10, num_iters_backward=None):
    if num_iters_backward is None:
        num_iters_backward = num_iters
    if num_iters_backward < 0:
        raise RuntimeError('num_iters_backward must not be negative')
    return _MatrixSquareRootNSLyap.apply(a, num_iters, num_iters_backward)


class _MatrixSquareRootEig(torch.autograd.Function):
    """Square root through the symmetric eigendecomposition (reference sqrtm.py:58-70); the backward solves the
    Sylvester equation in the eigenbasis: X_ij = (V^T G V)_ij / (s_i + s_j)."""

    @staticmethod
    def forward(ctx, a):
        vals, vecs = torch.linalg.eigh(a)
        ctx.save_for_backward(vals, vecs)
        return vecs @ vals.abs().sqrt().diag_embed() @ vecs.transpose(-2, -1)

    @staticmethod
    def backward(ctx, grad_output):
        vals, vecs = ctx.saved_tensors
        s = vals.abs().sqrt()
        denom = s.unsqueeze(-1) + s.unsqueeze(-2)
        vt = vecs.transpose(-2, -1)
        return vecs @ (vt @ grad_output @ vecs / denom) @ vt


def sqrtm_eig(a):
    _check_square(a)
    return _MatrixSquareRootEig.apply(a)
