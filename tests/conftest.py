"""pytest configuration: the `gpu` marker, import paths, shared helpers."""
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_ROOT = os.path.join(REPO, 'style-transfer-pytorch_amd')
for p in (PKG_ROOT, os.path.join(REPO, 'oracle'), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run through gpurun)')


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))


@pytest.fixture(scope='session')
def vgg_weights():
    from style_transfer import vgg
    params = vgg.synthetic_vgg19_weights(0)
    fp = np.load(os.path.join(GOLDEN, 'weights_fingerprint.npz'))['fp']
    got = np.array(vgg.weights_fingerprint(params))
    assert np.allclose(got, fp, rtol=1e-9, atol=1e-9), 'synthetic weight RNG drifted from the golden fixtures'
    return params


def rel_l2(a, b):
    a = torch.as_tensor(a, dtype=torch.float64).flatten()
    b = torch.as_tensor(b, dtype=torch.float64).flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))
