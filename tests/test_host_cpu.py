"""CPU-only checks of the host side: the C-ABI library loads and exports every declared symbol, struct mirrors match,
the drop-in modules keep the reference's API / state_dict contract, and the product refuses to run without the HIP
path (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

import simple3d_former_amd as s3d
from simple3d_former_amd import _lib as L
from simple3d_former_amd.engine import ParamArena, voxel_param_shapes
from oracle import voxel_oracle as vo


@pytest.fixture(scope='module')
def built():
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return L.lib()


def test_library_exports_every_declared_symbol(built):
    assert len(L.DECLARED_FUNCTIONS) >= 18
    for fn in L.DECLARED_FUNCTIONS:
        assert hasattr(built, fn), fn
    assert built.s3d_version() >= 100
    assert built.s3d_sizeof(b'NoSuchStruct') == 0
    for n, S in L.STRUCTS.items():
        assert built.s3d_sizeof(n.encode()) == ctypes.sizeof(S), n


def _gfx950_code_objects(path):
    """Every gfx950 ELF inside the library's .hip_fatbin section (uncompressed clang offload bundles, one per translation unit)."""
    import struct
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        fb = os.path.join(td, 'fb.bin')
        subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objcopy', f'--dump-section=.hip_fatbin={fb}', path, os.path.join(td, 'copy.so')], check=True)
        blob = open(fb, 'rb').read()
    magic, out, at = b'__CLANG_OFFLOAD_BUNDLE__', [], 0
    while (at := blob.find(magic, at)) >= 0:
        n = struct.unpack_from('<Q', blob, at + 24)[0]
        q = at + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if 'gfx950' in triple and size:
                out.append(blob[at + off:at + off + size])
        at += 24
    return out


def test_no_kernel_of_the_product_library_uses_scratch(built, tmp_path):
    """private segment = 0 for every kernel -- a spilling kernel is a design error here
    (512 registers per lane), and scratch traffic does not show in the algorithmic-bytes accounting of the roofline."""
    import subprocess
    if not os.path.exists('/opt/rocm/lib/llvm/bin/llvm-readelf'):
        pytest.skip('no llvm-readelf')
    objs = _gfx950_code_objects(L.LIB_PATH)
    assert len(objs) >= 6
    kernels, bad = 0, []
    for i, o in enumerate(objs):
        f = tmp_path / f'co{i}.elf'
        f.write_bytes(o)
        notes = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--notes', str(f)], capture_output=True, text=True, check=True).stdout
        name = None
        for ln in notes.splitlines():
            m = re.match(r'\s*\.name:\s+(\S+)', ln)
            if m and m.group(1).startswith('_Z'):
                name = m.group(1)
            m = re.match(r'\s*\.private_segment_fixed_size:\s+(\d+)', ln)
            if m:
                kernels += 1
                if int(m.group(1)):
                    bad.append((name, int(m.group(1))))
    assert kernels > 100, kernels
    assert not bad, bad


def test_header_parser_handles_every_struct():
    assert {'S3dGemmArgs', 'S3dAttnArgs', 'S3dBlockActs', 'S3dAdamState'} <= set(L.STRUCTS)
    f = dict(L.S3dGemmArgs._fields_)
    assert f['A_hi'] is ctypes.c_void_p and f['lda'] is ctypes.c_long and f['alpha'] is ctypes.c_float
    assert dict(L.S3dCeArgs._fields_)['target'] is ctypes.c_void_p
    assert ctypes.sizeof(L.S3dAdamState) == 36


def test_null_args_fail_loudly_not_abort(built):
    rc = built.s3d_gemm(0, 0, 0, 0, None, 1, None)
    assert rc != 0 and b'null' in built.s3d_last_error_string()
    with pytest.raises(RuntimeError):
        L.check(rc, 'gemm')


@pytest.mark.parametrize('kind,npatch', [('VoxelEmbed', 25), ('VoxelEmbed_no_average', 125), ('VoxelNaiveProjection', 25)])
def test_tokenizer_module_api(kind, npatch):
    m = getattr(s3d, kind)(voxel_size=30, cell_size=6, patch_size=5, embed_dim=384)
    assert m.voxel_size == (30, 30, 30) and m.cell_size == (6, 6, 6) and m.patch_size == 5
    assert m.num_patches == npatch and m.embed_dim == 384
    conv = 'conv2d_1' if kind == 'VoxelNaiveProjection' else 'conv3d_1'
    assert set(m.state_dict()) == {f'proj.{conv}.weight', f'proj.{conv}.bias'}
    with pytest.raises(AssertionError, match=r"Input voxel size \(32\*32\*32\) doesn't match model \(30\*30\*30\)"):
        m(torch.zeros(1, 1, 32, 32, 32))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(torch.zeros(1, 1, 30, 30, 30))


@pytest.mark.parametrize('head', ['default', 'AMSoftmax'])
def test_model_state_dict_contract_matches_reference_keys(head):
    kw = dict(backbone='deit_small_patch16_224', embed_layer='VoxelEmbed', voxel_size=30, cell=6, patch=5, n_classes=40, head=head)
    sd = vo.init_state_dict(**kw)          # key set verified against the reference (strict load) by make_golden.py
    model = s3d.Feature3D_ViT2D_V2(embed_layer=s3d.VoxelEmbed(voxel_size=30, cell_size=6, patch_size=5, embed_dim=384),
                                   n_classes=40, transformer_backbone='deit_small_patch16_224', pretrained=False,
                                   pos_embedding='default', head=head)
    model.load_state_dict(sd, strict=True)
    assert sum(p.numel() for p in model.parameters()) == (22159376 if head == 'default' else 22159336)
    blk = model.blocks[0]
    assert blk.attn.num_heads == 6 and abs(blk.attn.scale - 64 ** -0.5) < 1e-12 and blk.attn.qkv.weight.shape == (1152, 384)
    assert float(model.voxel_pos_embed.abs().sum()) == 0.0       # zeros, never random-initialised (reference quirk)
    assert model.norm.eps == 1e-6
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        model(torch.zeros(1, 1, 30, 30, 30))


def test_reference_error_conventions():
    te = s3d.VoxelEmbed(voxel_size=12, cell_size=4, patch_size=3, embed_dim=192)
    with pytest.raises(ValueError, match='Unknown transformer backbone name!'):
        s3d.Feature3D_ViT2D_V2(embed_layer=te, transformer_backbone='resnet50', pretrained=False)
    with pytest.raises(ValueError, match='Unknown positional embedding scheme!'):
        s3d.Feature3D_ViT2D_V2(embed_layer=te, transformer_backbone='deit_tiny_patch16_224', pretrained=False,
                               pos_embedding='bogus')
    with pytest.raises(RuntimeError, match='no network'):
        s3d.Feature3D_ViT2D_V2(embed_layer=te, transformer_backbone='deit_tiny_patch16_224', pretrained=True)
    m = s3d.Feature3D_ViT2D_V2(embed_layer=te, transformer_backbone='deit_base_patch16_224', pretrained=False)
    assert m.num_heads == 3 and m.embed_dim == 768                 # deit_base is built with 3 heads (reference quirk)


def test_param_arena_layout_and_used_parameter_set():
    kw = dict(backbone='deit_small_patch16_224', embed_layer='VoxelEmbed', cell=6, patch=5, n_classes=40)
    shapes = voxel_param_shapes(**kw)
    sd = vo.init_state_dict(voxel_size=30, **kw)
    assert set(shapes) == set(vo.used_param_names(sd))              # exactly the parameters the voxel forward touches
    assert sum(int(torch.tensor(s).prod()) for s in shapes.values()) == 21403432
    arena = ParamArena(shapes, torch.device('cpu'))
    offs = [arena.offsets[k] for k in shapes]
    assert offs == sorted(offs) and all(o % 8 == 0 for o in offs) and arena.numel % 8 == 0
    arena.load(sd)
    for k in shapes:
        assert torch.equal(arena.param(k), sd[k].reshape(shapes[k]))
    # blocks are laid out in forward order => gradient buckets complete back-to-front
    assert arena.offsets['blocks.0.norm1.weight'] < arena.offsets['blocks.11.mlp.fc2.bias'] < arena.offsets['norm.weight']


def test_engine_requires_the_gpu():
    with pytest.raises(RuntimeError, match='MI355X'):
        s3d.VoxelEngine(backbone='deit_tiny_patch16_224', embed_layer='VoxelEmbed', voxel_size=12, cell=4, patch=3,
                        n_classes=10, device='cpu')


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure; the shipped package must not reference it."""
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'simple3d-former_amd')
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), f
                assert 'timm_shim' not in src, f


def _point_cfg(task):
    import types
    n, d, c = (1024, 6, 40) if task == 'cls' else (2048, 22, 50)
    return types.SimpleNamespace(num_point=n, num_class=c, input_dim=d,
                                 model=types.SimpleNamespace(nblocks=4, nneighbor=16, transformer_dim=512, head='default',
                                                             transformer_backbone='deit_tiny_patch16_224', pretrained=False, name='3DViT'))


@pytest.mark.parametrize('task', ['cls', 'seg'])
def test_point_module_state_dict_contract(task):
    """Key names + shapes equal the reference's PointTransformerCls/Seg state_dict (fixture captured from the reference by
    tests/golden/make_golden_points.py tooling), incl. the parameters the reference creates but never uses."""
    import json
    from tests._util import GOLDEN
    from oracle import point_oracle as po
    ref = json.load(open(f'{GOLDEN}/point_state_dict_keys.json'))[task]
    cfg = _point_cfg(task)
    model = (s3d.PointTransformerCls if task == 'cls' else s3d.PointTransformerSeg)(cfg)
    assert cfg.embed_dim == 192                                        # written back like models/3DViT/model.py:221
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == ref
    # the engine's parameter set is exactly the set the oracle's forward touches
    sd = po.init_state_dict(backbone='deit_tiny_patch16_224', n_classes=cfg.num_class, d_points=cfg.input_dim)
    shapes = s3d.point_param_shapes('deit_tiny_patch16_224', cfg.num_class, cfg.input_dim)
    assert set(shapes) == set(po.used_param_names(sd))
    assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in shapes)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        model(torch.zeros(1, cfg.num_point, cfg.input_dim))
    cfg.model.transformer_backbone = 'resnet50'
    with pytest.raises(ValueError, match='Unknown transformer backbone name!'):
        s3d.PointTransformerCls(cfg)


@pytest.mark.parametrize('name', ['3DViT_1_layer', '3DViT_0_layer', '3DViT_LWF'])
def test_point_variant_modules_state_dict_contract(name):
    """model_module(name).PointTransformerSeg(cfg) == the reference's models/<name>/model.py class: state_dict keys + shapes
    (fixture from the reference), level plan, the engine's parameter set == what the oracle's forward + forward_images touch."""
    import json
    from tests._util import GOLDEN
    from oracle import point_oracle as po
    ref = json.load(open(f'{GOLDEN}/point_state_dict_keys.json'))[name]
    cfg = _point_cfg('seg')
    cfg.model.name = name
    model = s3d.model_module(name).PointTransformerSeg(cfg)
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == ref
    assert not hasattr(s3d.model_module(name), 'PointTransformerCls') and hasattr(s3d.model_module('3DViT'), 'PointTransformerCls')
    with pytest.raises(ModuleNotFoundError):
        s3d.model_module('3DViT_2_layer')
    sd = po.init_state_dict(backbone='deit_tiny_patch16_224', n_classes=50, d_points=22, variant=name)
    shapes = s3d.point_param_shapes('deit_tiny_patch16_224', 50, 22, name)
    assert set(shapes) == set(po.used_param_names(sd, name, images=True))
    assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in shapes)
    from simple3d_former_amd.point_engine import level_plan
    assert level_plan(name, 192, 2048) == po.level_plan(name, 192, 2048)
    assert level_plan('3DViT', 192, 2048) == (48, [2048, 512], [96, 192])
    assert level_plan('3DViT_LWF', 192, 2048) == (48, [512, 128], [96, 192])
    assert level_plan('3DViT_1_layer', 384, 2048) == (192, [512], [384])
    assert level_plan('3DViT_0_layer', 192, 2048) == (192, [], [])
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        model(torch.zeros(1, 2048, 22))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        model.forward_images(torch.zeros(1, 3, 224, 224))
    with pytest.raises(AttributeError):
        s3d.PointTransformerSeg(_point_cfg('seg')).forward_images(torch.zeros(1, 3, 224, 224))


# ---------------------------------------------------------------------------------------------- schedules (f1, second half)
def _load_schedules():
    from simple3d_former_amd import schedules
    return schedules


def test_voxel_lr_schedule_matches_torch_steplr_plus_per_epoch_warmup_dampening():
    """train_cls_voxel.py:195-198,293-294 replayed with torch's own StepLR on a real Adam optimizer; the warm-up is
    pytorch_warmup.UntunedLinearWarmup as the library defines it (dampen(): lr *= min(1, (step + 1) / int(2 / (1 - beta2))),
    called once by the constructor and then once per EPOCH by the reference)."""
    sch = _load_schedules()
    w = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([w], lr=0.05)
    scheduler = torch.optim.lr_scheduler.StepLR(opt, step_size=20, gamma=0.5)
    period, k = int(2.0 / (1.0 - 0.999)), [0]

    def dampen():
        for g in opt.param_groups:
            g['lr'] *= min(1.0, (k[0] + 1) / period)
        k[0] += 1
    dampen()                                                      # UntunedLinearWarmup.__init__
    import warnings
    for epoch in range(65):
        assert abs(opt.param_groups[0]['lr'] - sch.voxel_lr(epoch)) <= 1e-12 * 0.05, epoch
        opt.step()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')                       # the epoch argument of step() is deprecated, the reference uses it
            scheduler.step(scheduler.last_epoch + 1)
        dampen()
    assert period == 1999 and abs(sch.voxel_lr(0) - 0.05 / period) < 1e-18 and sch.voxel_lr(20, warmup=False) == 0.025


def test_point_schedules_match_the_reference_formulas():
    sch = _load_schedules()
    w = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([w], lr=0.01, momentum=0.9)
    scheduler = torch.optim.lr_scheduler.StepLR(opt, step_size=50, gamma=0.3)          # train_cls.py:93
    for epoch in range(120):
        assert abs(opt.param_groups[0]['lr'] - sch.point_cls_lr(epoch)) <= 1e-15, epoch
        opt.step(); scheduler.step()
    for epoch in (0, 19, 20, 45, 200, 400):                       # train_partseg.py:121-130 with config/partseg.yaml
        assert sch.partseg_lr(epoch) == max(0.05 * (0.5 ** (epoch // 20)), 1e-5)
        mom = 0.9 * (0.5 ** (epoch // 20))
        assert sch.partseg_bn_momentum(epoch) == (0.01 if mom < 0.01 else mom)

    class Eng:                                                    # EpochSchedule drives set_lr / set_bn_momentum
        def set_lr(self, v): self.lr = v
        def set_bn_momentum(self, v): self.bn = v
    e = Eng()
    s = sch.EpochSchedule.for_partseg(e)
    assert s.begin_epoch(40) == 0.0125 and e.lr == 0.0125 and e.bn == 0.225


@pytest.mark.parametrize('Bb,N,D,H,depth,cls_only', [(64, 26, 384, 6, 12, True), (8, 513, 192, 3, 2, False), (3, 15, 768, 3, 1, True)])
def test_block_workspace_layout_is_consistent(built, Bb, N, D, H, depth, cls_only):
    """s3d_block_workspace_bytes / _carve (host-side arithmetic only, so it runs without a GPU): every buffer lies inside the
    allocation, 256-byte aligned, no two distinct buffers overlap, the residual stream chains block to block, the forward-only low
    planes are shared, and the range to clear covers exactly the class-row buffers."""
    built.s3d_block_workspace_bytes.restype = ctypes.c_size_t
    sh = L.S3dBlockShape(Bb=Bb, N=N, D=D, H=H, hidden=4 * D, eps=1e-6, split=1, cls_only_block=depth if cls_only else 0, fuse=0)
    fwd_only = built.s3d_block_workspace_bytes(ctypes.byref(sh), depth, 0)
    total = built.s3d_block_workspace_bytes(ctypes.byref(sh), depth, 1)
    assert 0 < fwd_only < total
    base = 0x7f0000000000
    acts = (L.S3dBlockActs * depth)()
    sc = L.S3dBlockScratch()
    zo, zb = ctypes.c_size_t(0), ctypes.c_size_t(0)
    L.check(built.s3d_block_workspace_carve(ctypes.byref(sh), depth, 1, ctypes.c_void_p(base), ctypes.c_size_t(total), acts, ctypes.byref(sc),
                                            ctypes.byref(zo), ctypes.byref(zb)), 'carve')
    assert built.s3d_block_workspace_carve(ctypes.byref(sh), depth, 1, ctypes.c_void_p(base), ctypes.c_size_t(total - 1), acts, ctypes.byref(sc),
                                           ctypes.byref(zo), ctypes.byref(zb)) != 0                  # too small -> error, not a silent overrun
    M, Hd, BHN = Bb * N, 4 * D, Bb * H * N
    size = dict(x_in=4 * M * D, x_mid=4 * M * D, x_out=4 * M * D, mean1=4 * M, rstd1=4 * M, mean2=4 * M, rstd2=4 * M, lse=4 * BHN,
                xn1_hi=2 * M * D, xn1_lo=2 * M * D, qkv_hi=6 * M * D, qkv_lo=6 * M * D, att_hi=2 * M * D, att_lo=2 * M * D, xn2_hi=2 * M * D,
                xn2_lo=2 * M * D, hpre=2 * M * Hd, hact_hi=2 * M * Hd, hact_lo=2 * M * Hd)
    spans = {}
    for i in range(depth):
        assert acts[i].hpre_lo is None
        for f, nbytes in size.items():
            ptr = getattr(acts[i], f)
            assert ptr is not None and ptr % 256 == 0 and base <= ptr and ptr + nbytes <= base + total, (i, f)
            spans.setdefault(ptr, nbytes)
            assert spans[ptr] == nbytes
        if i + 1 < depth:
            assert acts[i].x_out == acts[i + 1].x_in
            for f in ('xn1_lo', 'qkv_lo', 'xn2_lo', 'hact_lo'):
                assert getattr(acts[i], f) == getattr(acts[i + 1], f)
            assert acts[i].att_lo != acts[i + 1].att_lo and acts[i].xn1_hi != acts[i + 1].xn1_hi
    ssize = dict(dxn=4 * M * D, dx_a=4 * M * D, dx_b=4 * M * D, dx_a_bf=2 * M * D, dx_b_bf=2 * M * D, dh=2 * M * Hd, dqkv=6 * M * D, datt=2 * M * D,
                 delta=4 * BHN, ln_partial=4 * 2 * depth * sc.ln_partial_blocks * 2 * D)
    if cls_only:
        ssize.update(dx_b_cls=4 * M * D, dx_b_bf_cls=2 * M * D, datt_cls=2 * M * D)
        assert zo.value == sc.dx_b_cls - base and zo.value + zb.value == total
    else:
        assert sc.dx_b_cls is None and zb.value == 0
    for f, nbytes in ssize.items():
        ptr = getattr(sc, f)
        assert ptr is not None and ptr % 256 == 0 and ptr >= base + fwd_only and ptr + nbytes <= base + total, f
        spans[ptr] = nbytes
    assert sc.dx_a_lo is None
    order = sorted(spans.items())
    for (p0, n0), (p1, _) in zip(order, order[1:]):
        assert p0 + n0 <= p1, 'overlapping buffers'
