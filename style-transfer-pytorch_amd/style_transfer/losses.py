"""The reference's loss / bookkeeping modules as importable ``nn.Module`` classes (API completeness, SURVEY.md
section 2 row 2).

``stylize()`` does not build a module graph: the closure of reference style_transfer.py:472-476 is one call into
libst_amd.so.  User code written against the reference can however import these names from
``style_transfer.style_transfer`` - ``ScaledMSELoss``, ``ContentLoss``, ``ContentLossMSE``, ``StyleLoss``,
``StyleLossW2``, ``TVLoss``, ``SumLoss``, ``Scale``, ``LayerApply``, ``eye_like`` (reference :93-234) - so they are
provided here as plain torch modules with the reference's constructor arguments, buffers, static helpers and
arithmetic.  They run wherever their tensors live; ``StyleLossW2`` reaches the library's Newton-Schulz operators
through ``sqrtm.sqrtm_ns_lyap`` when its matrices sit on a HIP device.
"""

from functools import partial

import torch
from torch import nn
from torch.nn import functional as F

from . import sqrtm


class ScaledMSELoss(nn.Module):
    """Sum of squared differences over the L1 norm of the difference (+ eps): an MSE whose gradient has an L1
    norm of about one (reference :93-106)."""

    def __init__(self, eps=1e-8):
        super().__init__()
        self.register_buffer('eps', torch.tensor(eps))

    def extra_repr(self):
        return f'eps={self.eps:g}'

    def forward(self, input, target):
        delta = input - target
        return delta.pow(2).sum() / delta.abs().sum().add(self.eps)


class _TargetLoss(nn.Module):
    """loss(transform(input), target) with the target held as a buffer."""

    def __init__(self, target, loss):
        super().__init__()
        self.register_buffer('target', target)
        self.loss = loss

    def forward(self, input):
        return self.loss(input, self.target)


class ContentLoss(_TargetLoss):
    """Reference :109-116."""

    def __init__(self, target, eps=1e-8):
        super().__init__(target, ScaledMSELoss(eps=eps))


class ContentLossMSE(_TargetLoss):
    """Reference :119-126 - the content term stylize() uses (in the library: content_mse_kernel)."""

    def __init__(self, target):
        super().__init__(target, nn.MSELoss())


class StyleLoss(_TargetLoss):
    """Gram-matrix style loss (reference :129-142); the Gram matrix is divided by the number of positions."""

    def __init__(self, target, eps=1e-8):
        super().__init__(target, ScaledMSELoss(eps=eps))

    @staticmethod
    def get_target(target):
        flat = target.flatten(-2)
        return flat @ flat.transpose(-2, -1) / flat.shape[-1]

    def forward(self, input):
        return self.loss(self.get_target(input), self.target)


def eye_like(x):
    return torch.eye(x.shape[-2], x.shape[-1], dtype=x.dtype, device=x.device).expand_as(x)


class StyleLossW2(nn.Module):
    """Wasserstein-2 distance between the Gaussians fitted to the input's and the target's features
    (reference :149-181; in the library: st_gram.hip + the Newton-Schulz chains)."""

    def __init__(self, target, eps=1e-4):
        super().__init__()
        self.sqrtm = partial(sqrtm.sqrtm_ns_lyap, num_iters=12)
        mean, srm = target
        cov = self.srm_to_cov(mean, srm) + eye_like(srm) * eps
        self.register_buffer('mean', mean)
        self.register_buffer('cov', cov)
        self.register_buffer('cov_sqrt', self.sqrtm(cov))
        self.register_buffer('eps', mean.new_tensor(eps))

    @staticmethod
    def get_target(target):
        """(mean, second raw moment) over the spatial positions - linear in the features' distribution, so targets
        of several style images can be blended."""
        positions = target.shape[-2] * target.shape[-1]
        mean = target.mean([-2, -1])
        srm = torch.einsum('...chw,...dhw->...cd', target, target) / positions
        return mean, srm

    @staticmethod
    def srm_to_cov(mean, srm):
        return srm - torch.einsum('...c,...d->...cd', mean, mean)

    def forward(self, input):
        mean, srm = self.get_target(input)
        cov = self.srm_to_cov(mean, srm) + eye_like(srm) * self.eps
        mean_term = torch.mean((mean - self.mean) ** 2)
        cross = self.sqrtm(self.cov_sqrt @ cov @ self.cov_sqrt)
        cov_term = torch.diagonal(self.cov + cov - 2 * cross, dim1=-2, dim2=-1).mean()
        return mean_term + cov_term


class TVLoss(nn.Module):
    """L2 total variation over a nine-point stencil (reference :184-195; in the library: tv_interior_kernel)."""

    def forward(self, input):
        x = F.pad(input, (1, 1, 1, 1), 'replicate')
        centre = x[..., 1:-1, 1:-1]
        right = (x[..., 1:-1, 2:] - centre).pow(2).mean() / 3
        down = (x[..., 2:, 1:-1] - centre).pow(2).mean() / 3
        diag_se = (x[..., 1:, 1:] - x[..., :-1, :-1]).pow(2).mean() / 12
        diag_sw = (x[..., 1:, :-1] - x[..., :-1, 1:]).pow(2).mean() / 12
        return 2 * (right + down + diag_se + diag_sw)


class SumLoss(nn.ModuleList):
    """Sum of the member losses evaluated on the same arguments, on the last one's device (reference :198-208)."""

    def __init__(self, losses, verbose=False):
        super().__init__(losses)
        self.verbose = verbose

    def forward(self, *args, **kwargs):
        values = [member(*args, **kwargs) for member in self]
        if self.verbose:
            for i, value in enumerate(values):
                print(f'({i}): {value.item():g}')
        home = values[-1].device
        return sum(value.to(home) for value in values)


class Scale(nn.Module):
    """module(...) * scale (reference :211-221)."""

    def __init__(self, module, scale):
        super().__init__()
        self.module = module
        self.register_buffer('scale', torch.tensor(scale))

    def extra_repr(self):
        return f'(scale): {self.scale.item():g}'

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs) * self.scale


class LayerApply(nn.Module):
    """module(features[layer]) (reference :224-234)."""

    def __init__(self, module, layer):
        super().__init__()
        self.module = module
        self.layer = layer

    def extra_repr(self):
        return f'(layer): {self.layer!r}'

    def forward(self, input):
        return self.module(input[self.layer])
