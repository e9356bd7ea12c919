"""TEST INFRASTRUCTURE ONLY -- CPU (PyTorch fp32) restatement of the Simple3D-Former point-cloud hot path
(PointTransformerCls / PointTransformerSeg).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.

Pure functions over a flat {state_dict key: tensor} mapping with the reference's key names.  What each function follows
(paths relative to /root/reference):
  square_distance / index_points     data/pointnet_util.py:22-50
  farthest_point_sample              data/pointnet_util.py:53-73  (the random start index, :65, is an explicit argument here)
  set_abstraction (TransitionDown)   models/3DViT/model.py:33-39 -> data/pointnet_util.py:220-244 -> sample_and_group(knn=True)
                                     :99-138 (kNN = full argsort, first nsample); Conv2d 1x1 + BatchNorm2d + ReLU x2, max over k.
                                     The second kNN on new_xyz (:233-235) is dead code and is not restated.
  transition_up                      models/3DViT/model.py:42-72 -> PointNetFeaturePropagation.forward data/pointnet_util.py:381-420
                                     with mlp=[] and points1=None (3-NN by full sort, weights 1/(d+1e-8) normalised)
  forward_features / forward         models/3DViT/model.py:297-337 (cls: mean over points then head) and :494-535 (seg: per-point head)
  variants (VARIANTS / level_plan)   models/3DViT_1_layer/model.py:216-249,294-319 (one level), models/3DViT_0_layer/model.py:216-239,
                                     284-307 (no level), models/3DViT_LWF/model.py:216-249,294-321 (two levels, N/4 and N/16 points);
                                     forward_images :323-337 of each; LwF loss train_partseg_lwf.py:207-228
  timm blocks                        oracle.voxel_oracle.run_blocks (timm==0.3.2, un-vendored)
  losses                             train_cls.py:70,119-121 (CrossEntropyLoss), train_partseg.py:143-150 (CE over B*N rows)
  metrics                            train_cls.py:22-41 (instance / class accuracy), train_partseg.py:172-220 (part IoU, mIoU)

PINNING: tests/golden/make_golden_points.py runs the reference's own models/3DViT/model.py + data/pointnet_util.py (on
oracle/timm_shim, `data` package stubbed because data/__init__.py imports missing modules) with this module's deterministic
parameters and records logits / loss / gradients / BatchNorm running statistics and the FPS start indices it drew;
tests/test_oracle_points.py replays them here -- four fixtures for models/3DViT/model.py and four for the PointTransformerSeg of
models/3DViT_1_layer, 3DViT_0_layer, 3DViT_LWF (incl. forward_images and the LwF gradient).
"""
import math

import torch
import torch.nn.functional as F

from . import voxel_oracle as vo

BACKBONES = vo.BACKBONES
NSAMPLE = 16           # config/model/3DViT.yaml: nneighbor 16
BN_EPS = 1e-5

# The four model directories a trainer can name (train_partseg.py:74 / train_partseg_lwf.py import models.<name>.model):
#   levels     number of TransitionDown / TransitionUp pairs around the transformer
#   first_div  td 0 keeps N / first_div points (models/3DViT/model.py:242 `npoints // 4 ** i` vs
#              models/3DViT_1_layer/model.py:231 `npoints // 4 ** (i + 1)`)
#   head       state_dict key of the per-point head: the variants keep timm's 2-D `head` (Linear(D, 1000)) for
#              forward_images and add `new_head` (3DViT_1_layer/model.py:218-221)
#   image      forward_images exists (3DViT_1_layer/model.py:323-337); the base model replaces patch_embed by PointEmbed
# Base width C0 = D / 2**levels (fc1 / fc_pos_embed / head input): D/4, D/4, D/2, D.
VARIANTS = {
    '3DViT': dict(levels=2, first_div=1, head='head', image=False),
    '3DViT_LWF': dict(levels=2, first_div=4, head='new_head', image=True),
    '3DViT_1_layer': dict(levels=1, first_div=4, head='new_head', image=True),
    '3DViT_0_layer': dict(levels=0, first_div=1, head='new_head', image=True),
}


def level_plan(variant, D, n_points):
    """-> (C0, [npoint of td i], [out channels of td i])."""
    v = VARIANTS[variant]
    C0 = D >> v['levels']
    S = [n_points // (v['first_div'] * 4 ** i) for i in range(v['levels'])]
    return C0, S, [C0 * 2 ** (i + 1) for i in range(v['levels'])]


# ----------------------------------------------------------------------------- geometry
def square_distance(src, dst):
    return torch.sum((src[:, :, None] - dst[:, None]) ** 2, dim=-1)


def index_points(points, idx):
    raw = idx.shape
    flat = idx.reshape(raw[0], -1)
    res = torch.gather(points, 1, flat[..., None].expand(-1, -1, points.size(-1)))
    return res.reshape(*raw, -1)


def farthest_point_sample(xyz, npoint, start):
    """start: [B] int64, the value the reference draws with torch.randint at pointnet_util.py:65."""
    B, N, _ = xyz.shape
    centroids = torch.zeros(B, npoint, dtype=torch.long)
    distance = torch.full((B, N), 1e10)
    farthest = start.clone()
    bi = torch.arange(B)
    for i in range(npoint):
        centroids[:, i] = farthest
        c = xyz[bi, farthest].view(B, 1, 3)
        distance = torch.min(distance, torch.sum((xyz - c) ** 2, -1))
        farthest = torch.max(distance, -1)[1]
    return centroids


def knn_indices(query, ref, k):
    return square_distance(query, ref).argsort()[:, :, :k]


def three_nn_weights(xyz1, xyz2):
    d, idx = square_distance(xyz1, xyz2).sort(dim=-1)
    d, idx = d[:, :, :3], idx[:, :, :3]
    r = 1.0 / (d + 1e-8)
    return idx, r / r.sum(dim=2, keepdim=True)


# ----------------------------------------------------------------------------- layers
def _bn(x, sd, pre, training, momentum, stats_out):
    """x: [rows, C] -- BatchNorm over rows (BatchNorm2d over (B, k, S) / BatchNorm1d over (B, N) are the same thing once
    the channel axis is last).  Returns the normalised tensor; new running statistics go to stats_out."""
    rm, rv = sd[pre + 'running_mean'].clone(), sd[pre + 'running_var'].clone()
    y = F.batch_norm(x, rm, rv, sd[pre + 'weight'], sd[pre + 'bias'], training=training, momentum=momentum, eps=BN_EPS)
    if training and stats_out is not None:
        stats_out[pre + 'running_mean'], stats_out[pre + 'running_var'] = rm, rv
    return y


def set_abstraction(xyz, points, sd, pre, npoint, start, training, momentum, stats_out):
    """TransitionDown: FPS -> kNN(16) -> [xyz_rel, feats] -> 2 x (1x1 conv + BN + ReLU) -> max over the k neighbours."""
    B, N, _ = xyz.shape
    fps_idx = farthest_point_sample(xyz, npoint, start)
    new_xyz = index_points(xyz, fps_idx)                                        # [B,S,3]
    idx = knn_indices(new_xyz, xyz, NSAMPLE)                                    # [B,S,k]
    g = torch.cat([index_points(xyz, idx) - new_xyz[:, :, None], index_points(points, idx)], dim=-1)   # [B,S,k,3+C]
    rows = g.reshape(B * npoint * NSAMPLE, -1)
    for j in range(2):
        w = sd[f'{pre}sa.mlp_convs.{j}.weight']
        rows = rows @ w.reshape(w.shape[0], -1).t() + sd[f'{pre}sa.mlp_convs.{j}.bias']
        rows = F.relu(_bn(rows, sd, f'{pre}sa.mlp_bns.{j}.', training, momentum, stats_out))
    return new_xyz, rows.reshape(B, npoint, NSAMPLE, -1).max(dim=2)[0], fps_idx, idx


def transition_up(xyz1, points1, xyz2, points2, sd, pre, training, momentum, stats_out):
    """feats1 = ReLU(BN(Linear(points1))) lives on xyz1 (coarse); feats2 likewise on xyz2 (fine);
    out = interp_3nn(feats1: xyz1 -> xyz2) + feats2."""
    B = xyz1.shape[0]

    def branch(p, name):
        r = p.reshape(-1, p.shape[-1]) @ sd[f'{pre}{name}.0.weight'].t() + sd[f'{pre}{name}.0.bias']
        return F.relu(_bn(r, sd, f'{pre}{name}.2.', training, momentum, stats_out)).reshape(B, p.shape[1], -1)

    f1, f2 = branch(points1, 'fc1'), branch(points2, 'fc2')
    idx, w = three_nn_weights(xyz2, xyz1)                                       # for every fine point: 3 coarse neighbours
    return (index_points(f1, idx) * w[..., None]).sum(dim=2) + f2


def mlp2(x, sd, pre):
    h = F.relu(x @ sd[pre + '0.weight'].t() + sd[pre + '0.bias'])
    return h @ sd[pre + '2.weight'].t() + sd[pre + '2.bias']


# ----------------------------------------------------------------------------- model
def forward_features(sd, x, *, backbone, starts, training=True, momentum=0.1, stats_out=None, bf16=False, variant='3DViT'):
    """x: [B,N,d_points] (xyz first); starts = one FPS start-index tensor [B] per TransitionDown.
    3DViT / 3DViT_LWF: models/3DViT/model.py:297-327; 3DViT_1_layer/model.py:294-319; 3DViT_0_layer/model.py:284-307."""
    cfg = BACKBONES[backbone]
    D, depth, H = cfg['embed_dim'], cfg['depth'], cfg['num_heads']
    B, N, _ = x.shape
    _, S, _ = level_plan(variant, D, N)
    xyz = x[..., :3]
    f = mlp2(x, sd, 'fc1.') + mlp2(xyz, sd, 'fc_pos_embed.')
    pyramid = [(xyz, f)]                                          # (coordinates, features) of every resolution, fine -> coarse
    for i, npoint in enumerate(S):
        cx, cp = pyramid[-1]
        nx, np_, _, _ = set_abstraction(cx, cp, sd, f'transition_downs.{i}.', npoint, starts[i], training, momentum, stats_out)
        pyramid.append((nx, np_))
    t = torch.cat((sd['cls_token'].expand(B, -1, -1), pyramid[-1][1]), dim=1)
    t = vo.run_blocks(t, sd, depth, H, bf16)[:, 1:]
    for j in range(len(S)):
        (cx, _), (fx, fp) = pyramid[len(S) - j], pyramid[len(S) - j - 1]
        t = transition_up(cx, t, fx, fp, sd, f'transition_ups.{j}.', training, momentum, stats_out)
    return t


def forward(sd, x, *, task, variant='3DViT', **kw):
    feats = forward_features(sd, x, variant=variant, **kw)
    if task == 'cls':
        feats = feats.mean(1)
    hk = VARIANTS[variant]['head']
    if hk + '.W' in sd:                   # cfg.model.head == 'AMSoftmax' (models/3DViT/model.py:230-231, 427-428)
        return am_softmax_rows(feats, sd[hk + '.W'])
    return feats @ sd[hk + '.weight'].t() + sd[hk + '.bias']


def am_softmax_rows(x, W, s=30.0):
    """AMSoftmaxLayer.forward, models/3DViT/model.py:134-142: `B, N, C = x.shape` -- a 3-D input, i.e. the per-point head of
    PointTransformerSeg (PointTransformerCls hands it the 2-D x.mean(1) and fails on that unpack in the reference itself)."""
    B, N, C = x.shape
    x2 = x.reshape(-1, C)
    xn = x2 / torch.norm(x2, p=2, dim=1, keepdim=True).clamp(min=1e-12)
    wn = W / torch.norm(W, p=2, dim=0, keepdim=True).clamp(min=1e-12)
    return (xn @ wn * s).view(B, N, -1)


def forward_images(sd, img, *, backbone, bf16=False):
    """PointTransformerSeg.forward_images of the variants (models/3DViT_1_layer/model.py:323-337): the same arithmetic as
    Feature3D_ViT2D_V2.forward_images -- timm PatchEmbed, cls concat, + pos_embed, the shared blocks, norm, 2-D `head`."""
    return vo.forward_images(sd, img, backbone=backbone, bf16=bf16)


def loss_fn(logits, target):
    return F.cross_entropy(logits.reshape(-1, logits.shape[-1]), target.reshape(-1))


IMAGE_ONLY = ('pos_embed', 'patch_embed.', 'head.')


def used_param_names(sd, variant='3DViT', images=False):
    """Parameters reached by the point forward (+ the 2-D stem / head when the LwF image loss is added)."""
    skip = ('pos_embed', 'patch_embed.') if VARIANTS[variant]['head'] == 'head' else IMAGE_ONLY
    return [k for k in sd if sd[k].dtype.is_floating_point and 'running_' not in k and 'last_pos_embed' not in k
            and (images or not k.startswith(skip))]


def loss_and_grads(sd, x, target, **kw):
    names = used_param_names(sd, kw.get('variant', '3DViT'))
    leaf = {k: (v.detach().clone().requires_grad_(k in names) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    stats = {}
    logits = forward(leaf, x, stats_out=stats, **kw)
    loss = loss_fn(logits, target)
    grads = torch.autograd.grad(loss, [leaf[k] for k in names], allow_unused=True)
    return logits.detach(), loss.detach(), {k: g for k, g in zip(names, grads) if g is not None}, stats


def lwf_loss_and_grads(sd, x, target, img, img_target, lambda_weight=0.1, **kw):
    """train_partseg_lwf.py:207-228: CE(seg_pred, target) + lambda * CE(forward_images(images), label_teacher), one backward."""
    names = used_param_names(sd, kw['variant'], images=True)
    leaf = {k: (v.detach().clone().requires_grad_(k in names) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    stats = {}
    logits = forward(leaf, x, stats_out=stats, **kw)
    li = forward_images(leaf, img, backbone=kw['backbone'])
    loss = loss_fn(logits, target) + lambda_weight * F.cross_entropy(li, img_target)
    grads = torch.autograd.grad(loss, [leaf[k] for k in names], allow_unused=True)
    return logits.detach(), li.detach(), loss.detach(), {k: g for k, g in zip(names, grads) if g is not None}, stats


def sgd_momentum_step(p, g, buf, lr=0.01, momentum=0.9, first=False):
    """torch.optim.SGD(lr, momentum) (train_cls.py:91): buf = g on the first step, else momentum*buf + g; p -= lr*buf."""
    if first:
        buf.copy_(g)
    else:
        buf.mul_(momentum).add_(g)
    p.add_(buf, alpha=-lr)


# ----------------------------------------------------------------------------- metrics
def cls_accuracy(logits, target, n_classes):
    """train_cls.py:22-41: instance accuracy and mean per-class accuracy."""
    pred = logits.argmax(1)
    inst = float((pred == target).float().mean())
    accs = [float((pred[target == c] == c).float().mean()) for c in target.unique().tolist()]
    return inst, sum(accs) / len(accs)


def part_iou(logits, target, seg_classes):
    """train_partseg.py:181-206: argmax restricted to the parts of the shape's own category, then per-shape mean part IoU.
    seg_classes: {category: [part ids]}.  Returns the list of per-shape IoUs and their categories."""
    seg_label_to_cat = {l: c for c, ls in seg_classes.items() for l in ls}
    out = []
    for i in range(logits.shape[0]):
        cat = seg_label_to_cat[int(target[i, 0])]
        parts = seg_classes[cat]
        pred = logits[i][:, parts].argmax(1) + parts[0]
        ious = []
        for l in parts:
            gt, pr = target[i] == l, pred == l
            ious.append(1.0 if (int(gt.sum()) == 0 and int(pr.sum()) == 0) else float((gt & pr).sum()) / float((gt | pr).sum()))
        out.append((cat, sum(ious) / len(ious)))
    return out


# ----------------------------------------------------------------------------- parameters / inputs
def init_state_dict(*, backbone, n_classes, d_points, seed=9, variant='3DViT', head='default'):
    """Deterministic (integer-hash) parameters with the reference's key names and shapes for every tensor the point
    forward touches (+ BatchNorm buffers).  Unused reference parameters (pos_embed, patch_embed.*, sa.last_pos_embed.*)
    are not generated."""
    cfg = BACKBONES[backbone]
    D, depth = cfg['embed_dim'], cfg['depth']
    vv = VARIANTS[variant]
    levels = vv['levels']
    C0 = D >> levels
    sid = [0]
    sd = {}

    def u(shape, bound):
        sid[0] += 1
        return ((vo.portable_uniform(shape, seed, 5000 + sid[0]) * 2 - 1) * bound).float()

    def lin(key, o, i):
        sd[key + '.weight'] = u((o, i), 1.0 / math.sqrt(i)); sd[key + '.bias'] = u((o,), 1.0 / math.sqrt(i))

    def bn(key, c):
        sd[key + '.weight'] = 1 + u((c,), 0.1); sd[key + '.bias'] = u((c,), 0.1)
        sd[key + '.running_mean'] = u((c,), 0.1); sd[key + '.running_var'] = 1 + u((c,), 0.2)
        sd[key + '.num_batches_tracked'] = torch.zeros((), dtype=torch.long)

    lin('fc1.0', C0, d_points); lin('fc1.2', C0, C0)
    lin('fc_pos_embed.0', C0, 3); lin('fc_pos_embed.2', C0, C0)
    for i in range(levels):
        ch = C0 * 2 ** (i + 1)
        cin = ch // 2 + 3
        p = f'transition_downs.{i}.sa.'
        sd[p + 'mlp_convs.0.weight'] = u((ch, cin, 1, 1), 1.0 / math.sqrt(cin)); sd[p + 'mlp_convs.0.bias'] = u((ch,), 1.0 / math.sqrt(cin))
        bn(p + 'mlp_bns.0', ch)
        sd[p + 'mlp_convs.1.weight'] = u((ch, ch, 1, 1), 1.0 / math.sqrt(ch)); sd[p + 'mlp_convs.1.bias'] = u((ch,), 1.0 / math.sqrt(ch))
        bn(p + 'mlp_bns.1', ch)
    sd['cls_token'] = u((1, 1, D), 0.02 * math.sqrt(3))
    for i in range(depth):
        p = f'blocks.{i}.'
        sd[p + 'norm1.weight'] = 1 + u((D,), 0.1); sd[p + 'norm1.bias'] = u((D,), 0.1)
        sd[p + 'attn.qkv.weight'] = u((3 * D, D), 0.02 * math.sqrt(3)); sd[p + 'attn.qkv.bias'] = u((3 * D,), 0.05)
        sd[p + 'attn.proj.weight'] = u((D, D), 0.02 * math.sqrt(3)); sd[p + 'attn.proj.bias'] = u((D,), 0.05)
        sd[p + 'norm2.weight'] = 1 + u((D,), 0.1); sd[p + 'norm2.bias'] = u((D,), 0.1)
        sd[p + 'mlp.fc1.weight'] = u((4 * D, D), 0.02 * math.sqrt(3)); sd[p + 'mlp.fc1.bias'] = u((4 * D,), 0.05)
        sd[p + 'mlp.fc2.weight'] = u((D, 4 * D), 0.02 * math.sqrt(3)); sd[p + 'mlp.fc2.bias'] = u((D,), 0.05)
    sd['norm.weight'] = 1 + u((D,), 0.1); sd['norm.bias'] = u((D,), 0.1)
    for j, i in enumerate(reversed(range(levels))):
        ch = C0 * 2 ** i
        p = f'transition_ups.{j}.'
        lin(p + 'fc1.0', ch, ch * 2); bn(p + 'fc1.2', ch)
        lin(p + 'fc2.0', ch, ch); bn(p + 'fc2.2', ch)
    if head == 'AMSoftmax':               # AMSoftmaxLayer.W [in_feats][n_classes], xavier_normal_ (models/3DViT/model.py:132-133)
        sd[vv['head'] + '.W'] = u((C0, n_classes), math.sqrt(6.0 / (C0 + n_classes)))
    else:
        lin(vv['head'], n_classes, C0)
    if vv['image']:                       # timm's 2-D stem / head, kept by the variants for forward_images
        sd['patch_embed.proj.weight'] = u((D, 3, 16, 16), 1.0 / math.sqrt(768)); sd['patch_embed.proj.bias'] = u((D,), 1.0 / math.sqrt(768))
        sd['pos_embed'] = u((1, 197, D), 0.02 * math.sqrt(3))
        lin('head', 1000, D)
    return sd


def synthetic_class_points(batch, n_points, labels, seed=9, variant='3DViT'):
    """A LEARNABLE portable synthetic classification batch (trained-state fixture, tests/golden/make_golden_points_trained.py): the
    label stretches the cloud along z -- the i-th entry of `labels` scales z by 0.15 + 0.85 i / (len - 1) before the unit-sphere projection --
    and tilts every normal towards +z by the same amount, so both the geometry (FPS / kNN neighbourhoods) and the per-point features carry
    it.  Same FPS start recipe as synthetic_points."""
    u = lambda shape, s: vo.portable_uniform(shape, seed, 7100 + s)
    labels = list(labels)
    pick = (u((batch,), 4) * len(labels)).long().clamp(max=len(labels) - 1)
    y = torch.tensor(labels, dtype=torch.long)[pick]
    a = (0.15 + 0.85 * pick.double() / max(1, len(labels) - 1)).view(batch, 1, 1)
    xyz = u((batch, n_points, 3), 1) * 2 - 1
    xyz = torch.cat([xyz[..., :2], xyz[..., 2:] * a], dim=-1)
    r = xyz.norm(dim=-1, keepdim=True)
    xyz = torch.where(r > 1, xyz / r, xyz)
    nrm = u((batch, n_points, 3), 2) * 2 - 1
    nrm = torch.cat([nrm[..., :2], nrm[..., 2:] + 2.0 * a], dim=-1)
    nrm = nrm / nrm.norm(dim=-1, keepdim=True).clamp(min=1e-9)
    x = torch.cat([xyz, nrm], dim=-1).float().contiguous()
    _, S, _ = level_plan(variant, 4, n_points)
    n_in = ([n_points] + S)[:len(S)]
    starts = tuple((u((batch,), 10 + i) * n).long().clamp(max=n - 1) for i, n in enumerate(n_in))
    return x, y, starts


def synthetic_points(batch, n_points, d_points, n_classes, task, seed=9, variant='3DViT'):
    """xyz uniform in the unit ball (cf. pc_normalize, pointnet_util.py:15-20), unit normals, for
    part-seg a one-hot(16) object label appended (train_partseg.py:143); targets randint.  Integer-hash generator."""
    u = lambda shape, s: vo.portable_uniform(shape, seed, 7000 + s)
    xyz = u((batch, n_points, 3), 1) * 2 - 1
    r = xyz.norm(dim=-1, keepdim=True)
    xyz = torch.where(r > 1, xyz / r, xyz)                       # project outliers onto the unit sphere
    nrm = u((batch, n_points, 3), 2) * 2 - 1
    nrm = nrm / nrm.norm(dim=-1, keepdim=True).clamp(min=1e-9)
    feats = [xyz, nrm]
    if d_points > 6:
        lab = (u((batch,), 3) * (d_points - 6)).long().clamp(max=d_points - 7)
        feats.append(F.one_hot(lab, d_points - 6).double()[:, None].expand(-1, n_points, -1))
    x = torch.cat(feats, dim=-1)[..., :d_points].float().contiguous()
    if task == 'cls':
        y = (u((batch,), 4) * n_classes).long().clamp(max=n_classes - 1)
    else:
        y = (u((batch, n_points), 4) * n_classes).long().clamp(max=n_classes - 1)
    _, S, _ = level_plan(variant, 4, n_points)
    n_in = ([n_points] + S)[:len(S)]                                 # FPS of level i draws its start among that level's INPUT points
    starts = tuple((u((batch,), 10 + i) * n).long().clamp(max=n - 1) for i, n in enumerate(n_in))
    return x, y, starts
