"""Coverage guard (runs last: pytest orders files alphabetically): every kernel instantiation that ONE training step of a benched
configuration launches -- cfg-2 at batch 64, cfg-3 at batch 16 and 64, cfg-4 at batch 128, cfg-5 at batch 32, in the default
(non-deterministic) dispatch bench.py times -- must also have been launched inside a test that compares the HIP path with the CPU
oracle, an fp64 reference of the same operator or a reference-generated golden fixture (tests/conftest.py: ORACLE_COMPARED records
those launches through s3d_cov_enable / s3d_cov_collect).  A shape-dependent dispatch (s3d_gemm_pick_tile, the 'long rows' rules,
the attention regimes) that only the benched size reaches can therefore not ship without an oracle comparison.

Keys are "family:instantiation" (GEMMs: tiles | transposes | split | epilogue | ring; attention / LayerNorm / BatchNorm: the
template choice).  The guard needs the rest of the GPU suite to have run in the same session; alone it skips."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import simple3d_former_amd as s3d
    from simple3d_former_amd import _lib as L
    from simple3d_former_amd.point_engine import PointEngine

from oracle import point_oracle as po
from oracle import voxel_oracle as vo
from tests import _cov as C

DEV = 'cuda'


def _record(fn):
    lib = L.lib()
    lib.s3d_cov_collect.restype = __import__('ctypes').c_long
    lib.s3d_cov_enable(1)
    try:
        fn()
        torch.cuda.synchronize()
    finally:
        lib.s3d_cov_enable(0)
    return C.collect(lib)


def _voxel_step(B, **kw):
    group = kw.get('pos_embedding') == 'group_embed'
    sd = vo.init_state_dict(seed=9, **kw)
    x, y = vo.synthetic_batch(B, kw['voxel_size'], kw['n_classes'], seed=9)
    eng = s3d.VoxelEngine(device=DEV, **kw)
    eng.load_state_dict(sd)
    if group:
        eng.set_dropout(0.1, seed=5)                      # bench.py --config cfg3 trains with the reference's dropout
    xd, yd = x.to(DEV), y.to(DEV)
    eng.train_step(xd, yd)                                # warm-up: workspaces, kernel attributes
    return _record(lambda: eng.train_step(xd, yd))


def _point_step(task, n_points, d_points, n_classes, B):
    sd = po.init_state_dict(backbone='deit_tiny_patch16_224', n_classes=n_classes, d_points=d_points, seed=9)
    x, y, starts = po.synthetic_points(B, n_points, d_points, n_classes, task, seed=9)
    eng = PointEngine(backbone='deit_tiny_patch16_224', n_points=n_points, d_points=d_points, n_classes=n_classes, task=task, device=DEV)
    eng.load_state_dict(sd)
    xd, yd, sts = x.to(DEV), y.to(DEV), tuple(s.to(DEV) for s in starts)
    eng.train_step(xd, yd, sts)
    return _record(lambda: eng.train_step(xd, yd, sts))


CASES = {
    'cfg2_b64': lambda: _voxel_step(64, backbone='deit_small_patch16_224', embed_layer='VoxelEmbed', voxel_size=32, cell=6, patch=5, n_classes=40),
    'cfg3_b16': lambda: _voxel_step(16, backbone='deit_base_patch16_224', embed_layer='VoxelEmbed_no_average', voxel_size=128, cell=9, patch=14,
                                    n_classes=55, pos_embedding='group_embed'),
    'cfg3_b64': lambda: _voxel_step(64, backbone='deit_base_patch16_224', embed_layer='VoxelEmbed_no_average', voxel_size=128, cell=9, patch=14,
                                    n_classes=55, pos_embedding='group_embed'),
    'cfg4_b128': lambda: _point_step('cls', 1024, 6, 40, 128),
    'cfg5_b32': lambda: _point_step('seg', 2048, 22, 50, 32),
}


@pytest.mark.parametrize('case', list(CASES))
def test_every_benched_kernel_instantiation_has_met_the_oracle(case):
    if len(C.COVERED) < 20:
        pytest.skip('needs the oracle-compared GPU tests of the same session (run the whole suite: pytest tests -m gpu)')
    assert not L.lib().s3d_get_deterministic()
    launched = CASES[case]()
    assert len(launched) >= 8, launched
    missing = sorted(k for k in launched if k not in C.COVERED)
    print(f'{case}: {len(launched)} kernel instantiations, {sum(launched.values())} launches per step; all covered: {not missing}')
    assert not missing, f'{case}: launched by the benched step but by no oracle-compared test: {missing}'
