"""Minimal stand-in for torchvision so the UNMODIFIED reference imports in the build container.

Test infrastructure only (used by tests/golden/make_golden.py); never shipped, never imported by
the package.  torchvision is not installed in this image and there is no network.
"""
from . import models, transforms  # noqa: F401
