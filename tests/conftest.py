import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'fast-semantic-segmentation_amd')
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a ROCm GPU (MI355X); run with -m gpu')


def pytest_collection_modifyitems(config, items):
    # GPU tests never run implicitly on a machine without a GPU
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU in this environment')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
