"""Evaluation metrics and the .binvox reader ("next" rows f2/f3): oracle pinned on the reference-generated fixtures
(tests/golden/make_golden_eval.py), host logic on CPU, device kernels through the C ABI on the GPU."""
import io
import os

import numpy as np
import pytest
import torch

import simple3d_former_amd as s3d
from simple3d_former_amd import binvox, metrics
from oracle import eval_oracle as eo

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
BV = np.load(os.path.join(GOLD, 'binvox_cases.npz'))
EV = np.load(os.path.join(GOLD, 'eval_cases.npz'))
BV_NAMES = sorted({k.split('/')[0] for k in BV.files})
TABLE = metrics.SHAPENET_PARTS


def _dense(name):
    dims = BV[name + '/dims']
    shape = (dims[0], dims[2], dims[1])
    return np.unpackbits(BV[name + '/dense_packed'])[:int(np.prod(dims))].astype(bool).reshape(shape)


def _batches(prefix):
    out, i = [], 0
    while f'{prefix}/logits{i}' in EV.files:
        out.append((EV[f'{prefix}/logits{i}'], EV[f'{prefix}/target{i}']))
        i += 1
    return out


# ------------------------------------------------------------------------------------------------------------ CPU: oracle
@pytest.mark.parametrize('name', BV_NAMES)
def test_oracle_binvox_matches_reference(name):
    dense, dims, tr, sc = eo.binvox_read(BV[name + '/file'].tobytes())
    assert np.array_equal(dense, _dense(name))
    assert list(dims) == list(BV[name + '/dims']) and np.allclose(tr, BV[name + '/translate']) and sc == float(BV[name + '/scale'])
    assert int(np.int32(dense).sum()) == int(BV[name + '/int32_sum'])


def test_oracle_cls_voxel_metrics_match_reference():
    b = _batches('clsvox')
    tc, cc, ct = 0, np.zeros(40), np.zeros(40)
    for lg, t in b:
        _, c, a, n = eo.cls_counts(lg, t, 40)
        tc, cc, ct = tc + c, cc + a, ct + n
    assert tc == int(EV['clsvox/total_correct']) and int(ct.sum()) == int(EV['clsvox/total_testset'])
    assert np.array_equal(cc, EV['clsvox/class_correct']) and np.array_equal(ct, EV['clsvox/class_total'])
    acc, _ = eo.cls_voxel_metrics(b, 40)
    assert acc == float(EV['clsvox/accuracy'])


def test_oracle_cls_points_metrics_match_reference():
    inst, cls = eo.cls_points_metrics(_batches('clspts'), 40)
    assert inst == pytest.approx(float(EV['clspts/instance_acc']), abs=1e-12)
    assert cls == pytest.approx(float(EV['clspts/class_acc']), abs=1e-12)


def test_oracle_partseg_matches_reference():
    b = _batches('partseg')
    seen, corr = np.zeros(50, dtype=np.int64), np.zeros(50, dtype=np.int64)
    per_cat = {n: [] for n, _, _ in TABLE}
    name_of = {f: n for n, f, _ in TABLE}
    for i, (lg, t) in enumerate(b):
        pred, c, s, k, iou, first = eo.partseg_batch(lg, t, TABLE, 50)
        assert np.array_equal(pred, EV[f'partseg/pred{i}'])
        seen, corr = seen + s, corr + k
        for v, f in zip(iou, first):
            per_cat[name_of[int(f)]].append(v)
    assert np.array_equal(seen, EV['partseg/total_seen_class']) and np.array_equal(corr, EV['partseg/total_correct_class'])
    for n in per_cat:
        assert np.array_equal(np.array(per_cat[n]), EV[f'partseg/shape_ious/{n}']), n
    m = eo.partseg_metrics(b, TABLE, 50)
    for k in ('accuracy', 'class_avg_accuracy', 'class_avg_iou', 'inctance_avg_iou'):
        assert m[k] == pytest.approx(float(EV['partseg/' + k]), abs=1e-12), k


def test_part_table_matches_reference():
    got = sorted((n, f, c) for n, f, c in TABLE)
    ref = sorted((str(n), int(f), int(c)) for n, (f, c) in zip(EV['partseg/table_names'], EV['partseg/table']))
    assert got == ref
    assert metrics.seg_classes_dict()['Motorbike'] == [30, 31, 32, 33, 34, 35]


# ------------------------------------------------------------------------------------------------------------ CPU: host reader
@pytest.mark.parametrize('name', BV_NAMES)
def test_host_binvox_reader(name):
    data = BV[name + '/file'].tobytes()
    dense, dims, tr, sc = binvox.read_dense(data)
    assert dense.dtype == bool and np.array_equal(dense, _dense(name))
    assert list(dims) == list(BV[name + '/dims']) and tr == list(BV[name + '/translate']) and sc == float(BV[name + '/scale'])
    d2, *_ = binvox.read_dense(io.BytesIO(data))                       # file objects too
    assert np.array_equal(d2, dense)
    if dense.size % 32 == 0:
        words, *_ = binvox.read_packed(data)
        assert words.dtype == np.dtype('<u4') and np.array_equal(words, eo.pack_bits(dense))


@pytest.mark.parametrize('name', [n for n in BV_NAMES if n != 'box16x8x4'])
def test_host_binvox_writer_is_byte_identical_to_reference(name):
    buf = io.BytesIO()
    binvox.write(_dense(name), buf, translate=[float(v) for v in BV[name + '/translate']], scale=float(BV[name + '/scale']))
    assert buf.getvalue() == BV[name + '/file'].tobytes()


def test_host_binvox_errors():
    with pytest.raises(IOError, match='Not a binvox file'):
        binvox.read_dense(b'#notbinvox 1\ndim 2 2 2\n')
    good = BV['rand32/file'].tobytes()
    with pytest.raises(IOError):
        binvox.read_dense(good[:-3])
    with pytest.raises(ValueError):
        binvox.pack_grid(np.zeros((3, 3, 3), bool))


def test_non_cubic_roundtrip():
    g = np.arange(16 * 8 * 4).reshape(16, 8, 4) % 3 == 0
    buf = io.BytesIO()
    binvox.write(g, buf)
    back, *_ = binvox.read_dense(buf.getvalue())
    assert np.array_equal(back, g)


# ------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize('name', ['rand32', 'sparse32', 'empty32', 'full32'])
def test_gpu_unpack_voxels(name):
    words, dims, *_ = binvox.read_packed(BV[name + '/file'].tobytes())
    B = 3
    allw = np.concatenate([words, words[::-1].copy(), words])
    out = binvox.unpack_to_device(allw, B, 32)
    assert out.shape == (B, 1, 32, 32, 32) and out.dtype == torch.float32
    want = _dense(name).astype(np.float32)
    got = out.cpu().numpy()
    assert np.array_equal(got[0, 0], want) and np.array_equal(got[2, 0], want)
    assert got[1].sum() == want.sum()
    with pytest.raises(ValueError):
        binvox.unpack_to_device(allw, 2, 32)


@pytest.mark.gpu
def test_gpu_unpack_feeds_the_tokenizer_identically():
    """bit-packed input path == the reference's int32 -> .float() path (train_cls_voxel.py:246) through the whole forward."""
    from oracle import voxel_oracle as vo
    kw = dict(backbone='deit_tiny_patch16_224', embed_layer='VoxelEmbed', voxel_size=32, cell=6, patch=5, n_classes=40)
    eng = s3d.VoxelEngine(**kw)
    eng.load_state_dict(vo.init_state_dict(seed=5, portable=True, **kw))
    grids = np.stack([_dense('rand32'), _dense('sparse32')])
    x_ref = torch.from_numpy(np.int32(grids)[:, None]).cuda().float()
    a = eng.forward(x_ref).clone()
    x = binvox.unpack_to_device(np.concatenate([binvox.pack_grid(g) for g in grids]), 2, 32)
    assert torch.equal(x, x_ref)
    assert torch.equal(eng.forward(x), a)


@pytest.mark.gpu
def test_gpu_cls_eval_matches_reference_fixture():
    ev = s3d.ClsEvaluator(40)
    for lg, t in _batches('clsvox'):
        pred = ev.update(torch.from_numpy(lg).cuda(), torch.from_numpy(t).cuda(), return_pred=True)
        assert np.array_equal(pred.cpu().numpy(), np.argmax(lg, 1))                 # ties -> first maximum
    r = ev.result()
    assert r['total'] == int(EV['clsvox/total_testset'])
    assert r['accuracy'] == float(EV['clsvox/accuracy'])
    assert np.array_equal(r['class_correct'], EV['clsvox/class_correct']) and np.array_equal(r['class_total'], EV['clsvox/class_total'])
    want = EV['clsvox/class_acc']
    got = r['class_correct'] / r['class_total'].astype(np.float64)
    assert np.allclose(got[~np.isnan(want)], want[~np.isnan(want)], atol=1e-7) and np.array_equal(np.isnan(got), np.isnan(want))


@pytest.mark.gpu
def test_gpu_cls_eval_padded_rows_and_batch_average():
    """padded logits rows (ld > C) as the head kernel writes them; per-batch results reproduce train_cls.py:22-41."""
    b = _batches('clspts')
    inst, accs = [], np.zeros((40, 2))
    for lg, t in b:
        pad = torch.full((lg.shape[0], 64), 1e9, device='cuda')
        pad[:, :40] = torch.from_numpy(lg).cuda()
        ev = s3d.ClsEvaluator(40)
        ev.update(pad, torch.from_numpy(t).cuda(), ld=64)
        r = ev.result()
        inst.append(r['accuracy'])
        seen = r['class_total'] > 0
        accs[seen, 0] += r['class_correct'][seen] / r['class_total'][seen].astype(np.float64)
        accs[seen, 1] += 1
    assert np.mean(inst) == pytest.approx(float(EV['clspts/instance_acc']), abs=1e-12)
    assert np.mean(accs[:, 0] / accs[:, 1]) == pytest.approx(float(EV['clspts/class_acc']), abs=1e-12)


@pytest.mark.gpu
def test_gpu_partseg_eval_matches_reference_fixture():
    ev = s3d.PartSegEvaluator(50)
    for i, (lg, t) in enumerate(_batches('partseg')):
        pred = ev.update(torch.from_numpy(lg).cuda(), torch.from_numpy(t).cuda(), return_pred=True)
        assert np.array_equal(pred.cpu().numpy(), EV[f'partseg/pred{i}'])
    r = ev.result()
    for k in ('accuracy', 'class_avg_accuracy', 'class_avg_iou', 'inctance_avg_iou'):
        assert r[k] == pytest.approx(float(EV['partseg/' + k]), abs=1e-12), k
    want = np.concatenate([EV[f'partseg/shape_ious/{n}'] for n, _, _ in sorted(TABLE)])
    assert sorted(r['shape_ious'].tolist()) == sorted(want.tolist())                # fp64, bit-exact per shape


@pytest.mark.gpu
def test_gpu_partseg_eval_large_random_vs_oracle():
    rng = np.random.default_rng(3)
    B, N, P = 16, 2048, 50
    t = np.zeros((B, N), dtype=np.int64)
    for i in range(B):
        _, f, c = TABLE[i % 16]
        t[i] = rng.integers(f, f + c, N)
    lg = rng.standard_normal((B, N, 64)).astype(np.float32)
    lg[..., :P] += 2.5 * (np.arange(P)[None, None] == t[..., None]) * (rng.random((B, N, 1)) < 0.8)
    ev = s3d.PartSegEvaluator(P)
    pred = ev.update(torch.from_numpy(lg).cuda(), torch.from_numpy(t).cuda(), ld=64, return_pred=True)
    opred, c, seen, corr, iou, first = eo.partseg_batch(lg[..., :P], t, TABLE, P)
    assert np.array_equal(pred.cpu().numpy(), opred)
    r = ev.result()
    assert np.array_equal(r['shape_ious'], iou)
    m = eo.partseg_metrics([(lg[..., :P], t)], TABLE, P)
    for k in m:
        assert r[k] == pytest.approx(m[k], abs=1e-12), k
