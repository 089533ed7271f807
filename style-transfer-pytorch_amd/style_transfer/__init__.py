"""Neural style transfer (https://arxiv.org/abs/1508.06576) - MI355X-native hot path.

Same public names as the reference package (reference ``style_transfer/__init__.py:5,8-9``):
``srgb_profile``, ``STIterate``, ``StyleTransfer``, ``WebInterface``; console script
``style_transfer`` -> ``style_transfer.cli:main`` (setup.py).  The per-iteration work runs in the HIP
library ``lib/libst_amd.so`` (see ``include/st_amd.h``); there is no CPU fallback.
"""

from . import sqrtm  # noqa: F401
from .style_transfer import EMA, STIterate, StyleTransfer, VGGFeatures  # noqa: F401

__all__ = ['STIterate', 'StyleTransfer', 'WebInterface', 'srgb_profile']


def __getattr__(name):
    if name == 'WebInterface':          # lazily: aiohttp is only needed for --web
        from .web_interface import WebInterface
        return WebInterface
    # The reference ships an ICC file; here the equivalent sRGB profile is produced by LittleCMS on demand.
    if name == 'srgb_profile':
        from PIL import ImageCms
        data = ImageCms.ImageCmsProfile(ImageCms.createProfile('sRGB')).tobytes()
        globals()['srgb_profile'] = data
        return data
    raise AttributeError(name)
