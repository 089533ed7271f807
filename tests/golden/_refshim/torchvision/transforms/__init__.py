import torch

from . import functional  # noqa: F401


class Normalize(torch.nn.Module):
    """(x - mean) / std per channel, as torchvision.transforms.Normalize does for float tensors."""

    def __init__(self, mean, std, inplace=False):
        super().__init__()
        self.mean, self.std = list(mean), list(std)

    def forward(self, tensor):
        mean = torch.as_tensor(self.mean, dtype=tensor.dtype, device=tensor.device)
        std = torch.as_tensor(self.std, dtype=tensor.dtype, device=tensor.device)
        return tensor.clone().sub_(mean[:, None, None]).div_(std[:, None, None])
