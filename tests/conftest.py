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


def pytest_sessionstart(session):
    """CPU-oracle threads of the GPU suite.  The GPU pool's hosts are 2 x EPYC 9575F (256 hardware threads); torch's default intra-op pool
    there runs the oracle's fp32 steps several times SLOWER than a small one -- measured on test_cfg5_full_size_parity + the cfg-3 batch-3 step
    (tools/r6/run_threads_probe.sh, profiles/r06_gpu_tests.txt): 8 threads 21.9 s, **16 threads 16.8 s**, 32 threads 18.3 s, 64 threads 28.3 s,
    every hardware thread 99.1 s.  Round 5's suite spent two thirds of its 988 s (driver's box) there.  S3D_TEST_THREADS overrides; hosts with
    <= 32 threads keep torch's default."""
    try:
        import torch
    except Exception:
        return
    want = os.environ.get('S3D_TEST_THREADS')
    cores = os.cpu_count() or 8
    if want:
        torch.set_num_threads(max(1, min(int(want), cores)))
    elif cores > 32:
        torch.set_num_threads(16)


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
