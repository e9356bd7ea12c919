"""Golden-vector generator for the evaluation metrics and the .binvox reader (run ONLY in the build container, where
/root/reference exists):

    python tests/golden/make_golden_eval.py

* binvox: imports the reference's utils/binvox_rw.py (with the numpy-1 aliases np.bool / np.int it needs restored for the
  duration), writes synthetic grids with ITS writer and decodes them with ITS read_as_3d_array; the file bytes and the decoded
  grid (bit-packed) are the fixture.
* metrics: the reference's test loops are inline in its training scripts, so their source lines are read from
  /root/reference at generation time, dedented and executed on seeded logits / targets (nothing of that text is stored):
  train_cls_voxel.py:300-326, train_cls.py:22-41 (def test), train_partseg.py:26-33 (part table) + 178-220."""
import io
import os
import sys
import textwrap
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference'

from oracle import voxel_oracle as vo  # noqa: E402


def ref_lines(path, a, b):
    with open(os.path.join(REF, path)) as f:
        return textwrap.dedent(''.join(f.readlines()[a - 1:b]))


class numpy1_aliases:
    """Restores the numpy-1 aliases (np.bool / np.int / np.float) the reference uses, only where numpy lacks them."""

    def __enter__(self):
        self.added = [n for n in ('bool', 'int', 'float') if n not in np.__dict__ and not _has(n)]
        for n in self.added:
            setattr(np, n, {'bool': bool, 'int': int, 'float': float}[n])

    def __exit__(self, *a):
        for n in self.added:
            delattr(np, n)


def _has(n):
    try:
        getattr(np, n)
        return True
    except AttributeError:
        return False


def uniform(shape, seed, stream):
    return vo.portable_uniform(shape, seed, stream).numpy().astype(np.float64)


# ------------------------------------------------------------------------------------------------------------------ binvox
def binvox_cases():
    with numpy1_aliases():                                   # the reference predates numpy 1.24
        sys.path.insert(0, os.path.join(REF, 'utils'))
        import binvox_rw
        out = {}
        grids = {
            'rand32': uniform((32, 32, 32), 3, 1) < 0.3,
            'sparse32': uniform((32, 32, 32), 3, 2) < 0.01,
            'empty32': np.zeros((32, 32, 32), bool),
            'full32': np.ones((32, 32, 32), bool),             # runs longer than 255 must be split
            'slab30': (np.arange(30)[:, None, None] + np.arange(30)[None, :, None] * 2 + np.arange(30)[None, None, :] * 3) % 11 < 4,
            'box16x8x4': uniform((16, 8, 4), 3, 3) < 0.5,      # non-cubic: exercises the x-z-y order
        }
        for name, g in grids.items():
            # the reference writer expects its own storage order: Voxels.data [x,y,z] with axis_order 'xyz'
            vox = binvox_rw.Voxels(g.copy(), list(g.shape), [0.5, -1.25, 2.0], 0.75, 'xyz')
            buf = io.BytesIO()
            fp = types.SimpleNamespace(write=lambda s, buf=buf: buf.write(s.encode() if isinstance(s, str) else s))
            binvox_rw.write(vox, fp)
            data = buf.getvalue()
            back = binvox_rw.read_as_3d_array(io.BytesIO(data))
            if len(set(g.shape)) == 1:                       # the reference only round-trips cubic grids (reshape(dims) of the
                assert np.array_equal(back.data, g), name    # x-z-y payload); the non-cubic case pins ITS decode of ITS bytes
            out[name + '/file'] = np.frombuffer(data, dtype=np.uint8)
            out[name + '/dense_packed'] = np.packbits(back.data.reshape(-1))
            out[name + '/dims'] = np.array(back.dims)
            out[name + '/translate'] = np.array(back.translate)
            out[name + '/scale'] = np.array(back.scale)
            out[name + '/int32_sum'] = np.array(np.int32(back.data).sum())      # data/modelnet40.py:40
        np.savez_compressed(os.path.join(HERE, 'binvox_cases.npz'), **out)
        print('binvox_cases.npz', {k: v.shape for k, v in out.items() if k.endswith('/file')})


# ------------------------------------------------------------------------------------------------------------------ metrics
def make_logits(shape, target, n_classes, seed, stream, p_right=0.6):
    """Seeded logits whose argmax equals the target with probability ~p_right, with deliberate exact ties."""
    lg = (uniform(shape + (n_classes,), seed, stream) * 4 - 2).astype(np.float32)
    right = uniform(shape, seed, stream + 1) < p_right
    boost = np.zeros_like(lg)
    np.put_along_axis(boost, target[..., None], 3.0, axis=-1)
    lg = lg + boost * right[..., None].astype(np.float32)
    flat = lg.reshape(-1, n_classes)
    for r in range(0, flat.shape[0], 7):                     # ties: first maximum must win
        j = int(np.argmax(flat[r]))
        flat[r, (j + 3) % n_classes] = flat[r, j]
    return flat.reshape(lg.shape)


def cls_voxel_case(out):
    C = 40
    batches = []
    for b, n in enumerate((8, 8, 5)):
        t = (uniform((n,), 11, 10 + b) * 12).astype(np.int64) * 3 % C           # only some classes occur -> nan in class_acc
        batches.append((make_logits((n,), t, C, 11, 20 + 2 * b), t))
    src = ref_lines('train_cls_voxel.py', 300, 326)

    class FakeModel:
        def __init__(self):
            self.i = 0

        def eval(self):
            return self

        def __call__(self, voxel):
            lg = torch.from_numpy(batches[self.i][0])
            self.i += 1
            return lg

    ns = dict(torch=torch, np=np, tqdm=lambda it, total=None: it, N_CLASSES=C, device='cpu', model=FakeModel(),
              test_dataloader=[{'voxel': torch.zeros(len(t), 1), 'cls_idx': torch.from_numpy(t)} for _, t in batches])
    exec(src, ns)
    for i, (lg, t) in enumerate(batches):
        out[f'clsvox/logits{i}'], out[f'clsvox/target{i}'] = lg, t
    out['clsvox/total_correct'] = np.array(ns['total_correct'])
    out['clsvox/total_testset'] = np.array(ns['total_testset'])
    out['clsvox/class_correct'] = ns['class_correct'].numpy()
    out['clsvox/class_total'] = ns['class_total'].numpy()
    out['clsvox/class_acc'] = ns['class_acc'].numpy()
    out['clsvox/accuracy'] = np.array(ns['total_correct'] / float(ns['total_testset']))


def cls_points_case(out):
    C = 40
    batches = []
    for b, n in enumerate((16, 16, 9)):
        t = (uniform((n,), 12, 10 + b) * C).astype(np.int64)
        batches.append((make_logits((n,), t, C, 12, 20 + 2 * b), t))
    ns = dict(torch=torch, np=np, tqdm=lambda it, total=None: it)
    exec(ref_lines('train_cls.py', 22, 41), ns)

    class FakeModel:
        i = 0

        def eval(self):
            return self

        def __call__(self, pts):
            lg = torch.from_numpy(batches[FakeModel.i][0])
            FakeModel.i += 1
            return lg

    cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        # every class must occur for the reference's mean to be finite; append a batch that holds them all
        t_all = np.arange(C, dtype=np.int64)
        batches.append((make_logits((C,), t_all, C, 12, 40), t_all))
        loader = [(torch.zeros(len(t), 4, 3), torch.from_numpy(t)[:, None]) for _, t in batches]
        inst, cls = ns['test'](FakeModel(), loader, num_class=C)
    finally:
        torch.Tensor.cuda = cuda
    for i, (lg, t) in enumerate(batches):
        out[f'clspts/logits{i}'], out[f'clspts/target{i}'] = lg, t
    out['clspts/instance_acc'], out['clspts/class_acc'] = np.array(inst), np.array(cls)


def partseg_case(out):
    ns = dict(np=np)
    exec(ref_lines('train_partseg.py', 26, 33), ns)
    seg_classes = ns['seg_classes']
    P, N = 50, 96
    cats = sorted(seg_classes)
    batches = []
    for b, B in enumerate((7, 7, 6)):                         # 20 shapes: every category once, four of them twice
        t = np.zeros((B, N), dtype=np.int64)
        for i in range(B):
            parts = seg_classes[cats[(b * 7 + i) % 16]]
            present = parts[:len(parts) - (1 if b * 7 + i >= 16 else 0)]   # second visits lack a part (the IoU := 1.0 rule)
            t[i] = np.array(present)[(uniform((N,), 13, 100 * b + i) * len(present)).astype(np.int64)]
        lg = make_logits((B, N), t, P, 13, 50 + 2 * b, p_right=0.7)
        lg[:, :, :] += (uniform((B, N, P), 13, 60 + b) < 0.05).astype(np.float32) * 6      # out-of-category maxima to mask away
        batches.append((lg, t))
    with numpy1_aliases():                                   # train_partseg.py:216 uses np.float
        body = ref_lines('train_partseg.py', 178, 206)
        tail = ref_lines('train_partseg.py', 208, 220)
        st = dict(np=np, seg_classes=seg_classes, seg_label_to_cat=ns['seg_label_to_cat'], num_part=P, total_correct=0,
                  total_seen=0, total_seen_class=[0] * P, total_correct_class=[0] * P,
                  shape_ious={c: [] for c in seg_classes}, test_metrics={},
                  logger=types.SimpleNamespace(info=lambda *a: None))
        preds = []
        for lg, t in batches:
            st.update(cur_batch_size=t.shape[0], NUM_POINT=N, cur_pred_val_logits=lg,
                      target=types.SimpleNamespace(cpu=lambda t=t: types.SimpleNamespace(data=types.SimpleNamespace(numpy=lambda: t))))
            exec(body, st)
            preds.append(st['cur_pred_val'].copy())
        per_shape = {c: list(v) for c, v in st['shape_ious'].items()}
        exec(tail, st)
    for i, (lg, t) in enumerate(batches):
        out[f'partseg/logits{i}'], out[f'partseg/target{i}'], out[f'partseg/pred{i}'] = lg, t, preds[i]
    out['partseg/total_seen_class'] = np.array(st['total_seen_class'])
    out['partseg/total_correct_class'] = np.array(st['total_correct_class'])
    out['partseg/total_correct'] = np.array(st['total_correct'])
    for c in cats:
        out[f'partseg/shape_ious/{c}'] = np.array(per_shape[c])
    for k in ('accuracy', 'class_avg_accuracy', 'class_avg_iou', 'inctance_avg_iou'):
        out['partseg/' + k] = np.array(st['test_metrics'][k])
    out['partseg/table'] = np.array([[seg_classes[c][0], len(seg_classes[c])] for c in cats])
    out['partseg/table_names'] = np.array(cats)
    for c in cats:
        assert seg_classes[c] == list(range(seg_classes[c][0], seg_classes[c][0] + len(seg_classes[c])))


if __name__ == '__main__':
    binvox_cases()
    out = {}
    cls_voxel_case(out)
    cls_points_case(out)
    partseg_case(out)
    np.savez_compressed(os.path.join(HERE, 'eval_cases.npz'), **out)
    print('eval_cases.npz', {k: (v.item() if v.ndim == 0 else v.shape) for k, v in out.items() if 'logits' not in k and 'target' not in k and 'pred' not in k})
