"""Stub: reference cli.py imports tifffile at module level; nothing in the hot path uses it."""


class TIFF:
    pass


class TiffWriter:
    def __init__(self, *a, **k):
        raise RuntimeError('tifffile is not available in this image')
