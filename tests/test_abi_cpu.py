"""CPU checks of the drop-in boundary: the C-ABI library loads and exports exactly the symbols that
include/st_amd.h declares (no compute calls - there is no GPU here), and the product path refuses to
run without a GPU instead of falling back."""
import os
import re

import pytest

from conftest import REPO


def _header_symbols():
    text = open(os.path.join(REPO, 'include', 'st_amd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(st_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from style_transfer import _hip
    if not os.path.exists(_hip.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _hip.load_library(require_gpu=False)
    declared = _header_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in st_amd.h but not exported by libst_amd.so'
    assert sorted(_hip.EXPORTED_SYMBOLS) == declared, 'ctypes binding and header disagree'
    assert lib.st_abi_version() == 2
    assert lib.st_compiled_arch() == b'gfx950'


def test_no_cpu_fallback():
    import torch
    import style_transfer
    from style_transfer import _hip
    with pytest.raises(ValueError):
        style_transfer.StyleTransfer(devices=['cpu'], weights='synthetic')
    if not torch.cuda.is_available():
        with pytest.raises(_hip.HipLibraryError):
            _hip.load_library(require_gpu=True)


def test_package_does_not_import_the_oracle():
    pkg = os.path.join(REPO, 'style-transfer-pytorch_amd')
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(root, f)).read()
                assert 'st_oracle' not in src and 'oracle/' not in src, f'{f} references the oracle'
