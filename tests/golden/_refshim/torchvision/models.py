"""vgg19 (cfg "E") skeleton with the layer indexing of torchvision.models.vgg19().features."""
import enum

from torch import nn

_CFG_E = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M',
          512, 512, 512, 512, 'M']


class VGG19_Weights(enum.Enum):
    IMAGENET1K_V1 = 'imagenet1k_v1'
    DEFAULT = 'imagenet1k_v1'


class _VGG(nn.Module):
    def __init__(self):
        super().__init__()
        layers, cin = [], 3
        for v in _CFG_E:
            if v == 'M':
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)


def vgg19(weights=None, **kwargs):
    """Random-init VGG-19; the golden generator overwrites the conv parameters with seeded ones."""
    return _VGG()
