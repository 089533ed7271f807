import numpy as np
import torch
from PIL import Image


def to_tensor(pic):
    arr = np.asarray(pic, dtype=np.uint8)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    return torch.from_numpy(arr.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)


def to_pil_image(pic):
    arr = pic.detach().cpu().mul(255).byte().permute(1, 2, 0).numpy()
    return Image.fromarray(arr, 'RGB')
