import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tests import _cov  # noqa: E402  (one shared registry of oracle-compared launches)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


@pytest.fixture(autouse=True)
def _oracle_launch_coverage(request):
    name = getattr(request.node, 'originalname', None) or request.node.name
    if name not in _cov.ORACLE_COMPARED or request.node.get_closest_marker('gpu') is None:
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    from simple3d_former_amd import _lib as L
    lib = L.lib()
    lib.s3d_cov_collect.restype = __import__('ctypes').c_long
    lib.s3d_cov_enable(1)
    try:
        yield
    finally:
        lib.s3d_cov_enable(0)
        for k in _cov.collect(lib):
            _cov.COVERED.setdefault(k, set()).add(request.node.name)
