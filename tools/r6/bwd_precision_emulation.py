"""CPU emulation of the HIP backward's operand rounding on the stable-regime fixture (build container or anywhere: needs only oracle/).
Forward fp32 (the split-bf16 forward is fp32-class).  Backward per Linear layer y = x W^T + b:
    dgrad dx = Rd(dy) Rd(W),   wgrad dW = Rw(dy)^T Rw(x),   R = bf16 rounding ('b') or 16-mantissa-bit hi + lo ('s')
attention backward on bf16 q / k / v / dO / dS / P ('b') or unrounded ('s').  Modes: dgrad/wgrad/attn letters, e.g. bbb = the benched
default, sss = precise_backward, bsb = split wgrad only, sbs = split dgrad (+ attention) only.
    python tools/r6/bwd_precision_emulation.py bbb,bsb,sbs,sss [seeds]"""
import json, os, sys, time
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import voxel_oracle as vo
DEV = os.environ.get('DEVICE', 'cpu')      # 'cuda': the same torch arithmetic on the GPU box (fp32 matmuls; seconds per run)


def rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


def rs(t):
    hi = rb(t)
    return hi + rb(t - hi)


R = {'b': rb, 's': rs, 'x': lambda t: t}


class Lin(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, mode):
        ctx.save_for_backward(x, w)
        ctx.mode = mode
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        rd, rw = R[ctx.mode[0]], R[ctx.mode[1]]
        g2, x2 = gy.reshape(-1, gy.shape[-1]), x.reshape(-1, x.shape[-1])
        dx = (rd(g2) @ rd(w)).reshape(x.shape)
        dw = rw(g2).t() @ rw(x2)
        return dx, dw, g2.sum(0), None


class Attn(torch.autograd.Function):
    """softmax(q k^T scale) v with the backward of csrc/fused_block.hip: blk_attn_bwd_kernel (P recomputed from the stored operands)."""
    @staticmethod
    def forward(ctx, q, k, v, scale, mode):
        p = ((q @ k.transpose(-2, -1)) * scale).softmax(-1)
        ctx.save_for_backward(q, k, v)
        ctx.scale, ctx.mode = scale, mode
        return p @ v

    @staticmethod
    def backward(ctx, go):
        q, k, v = ctx.saved_tensors
        r = R[ctx.mode[2]]
        q, k, v, go = r(q), r(k), r(v), r(go)
        p = ((q @ k.transpose(-2, -1)) * ctx.scale).softmax(-1)
        dv = r(p).transpose(-2, -1) @ go
        dp = go @ v.transpose(-2, -1)
        ds = p * (dp - (p * dp).sum(-1, keepdim=True))
        ds = r(ds)
        return ds @ k * ctx.scale, ds.transpose(-2, -1) @ q * ctx.scale, dv, None, None


def forward(sd, x, mode, cell=6, H=6, depth=12):
    w, b = sd['voxel_embed.proj.conv3d_1.weight'], sd['voxel_embed.proj.conv3d_1.bias']
    t = vo.voxel_embed(x, w, b, cell).flatten(2).transpose(1, 2)
    B = x.shape[0]
    t = torch.cat((sd['cls_token'].expand(B, -1, -1), t), dim=1) + sd['voxel_pos_embed']
    D = t.shape[-1]
    hd = D // H
    for i in range(depth):
        p = f'blocks.{i}.'
        xn = vo.layer_norm(t, sd[p + 'norm1.weight'], sd[p + 'norm1.bias'])
        qkv = Lin.apply(xn, sd[p + 'attn.qkv.weight'], sd[p + 'attn.qkv.bias'], mode)
        N = t.shape[1]
        qkv = qkv.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
        a = Attn.apply(qkv[0], qkv[1], qkv[2], hd ** -0.5, mode).transpose(1, 2).reshape(B, N, D)
        t = t + Lin.apply(a, sd[p + 'attn.proj.weight'], sd[p + 'attn.proj.bias'], mode)
        xn = vo.layer_norm(t, sd[p + 'norm2.weight'], sd[p + 'norm2.bias'])
        h = F.gelu(Lin.apply(xn, sd[p + 'mlp.fc1.weight'], sd[p + 'mlp.fc1.bias'], mode))
        t = t + Lin.apply(h, sd[p + 'mlp.fc2.weight'], sd[p + 'mlp.fc2.bias'], mode)
    f = vo.layer_norm(t, sd['norm.weight'], sd['norm.bias'])[:, 0]
    return f @ sd['voxel_head.weight'].t() + sd['voxel_head.bias']


def run(cfg, seed, mode):
    kw = {k: cfg[k] for k in ('backbone', 'embed_layer', 'voxel_size', 'cell', 'patch', 'n_classes', 'pos_embedding', 'head')}
    sd = vo.init_state_dict(seed=seed, exercise_all=False, portable=True, **kw)
    names = vo.used_param_names(sd)
    P = {k: sd[k].clone().to(DEV).requires_grad_(True) for k in names}
    opt = torch.optim.Adam([P[k] for k in names], lr=cfg['lr'])
    dk = dict(base=cfg['density_base'], step=cfg['density_step'])
    data = [vo.synthetic_class_batch(cfg['batch'], cfg['voxel_size'], cfg['cell'], cfg['labels'], seed=500 + i, **dk) for i in range(cfg['n_batches'])]
    data = [(x.to(DEV), y.to(DEV)) for x, y in data]
    xh, yh = vo.synthetic_class_batch(cfg['held_batch'], cfg['voxel_size'], cfg['cell'], cfg['labels'], seed=999, **dk)
    xh, yh = xh.to(DEV), yh.to(DEV)
    losses, accs = [], []
    for step in range(cfg['steps']):
        x, y = data[step % len(data)]
        opt.zero_grad()
        loss = F.cross_entropy(forward(P, x, mode), y)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        if step + 1 in cfg['checkpoints']:
            with torch.no_grad():
                accs.append(float((forward(P, xh, 'xxx').argmax(1) == yh).float().mean()))
    return np.array(losses), np.array(accs)


def main():
    torch.set_num_threads(int(os.environ.get('THREADS', '8')))
    torch.backends.cuda.matmul.allow_tf32 = False
    modes = (sys.argv[1] if len(sys.argv) > 1 else 'bbb,bsb,sbs,sss').split(',')
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'trained_stable_cfg1_small_v30_adam400.npz'))
    cfg = json.loads(str(z['cfg']))
    if os.environ.get('FIXTURE') == 'adam60':      # round 5's unstable-regime fixture (Adam lr 1e-3 from the initialisation, 60 steps), more seeds
        cfg.update(steps=60, lr=1e-3, n_batches=4, labels=cfg['labels'][:9], density_base=0.04, density_step=0.07, held_batch=256, checkpoints=[40, 50, 60], tail=10)
    seeds = [int(s) for s in sys.argv[2].split(',')] if len(sys.argv) > 2 else cfg['seeds']
    if os.environ.get('SEEDS'):
        a, b = os.environ['SEEDS'].split('-')
        seeds = list(range(int(a), int(b) + 1))
    T = cfg['tail']
    for mode in modes:
        accs, tails = [], []
        for s in seeds:
            t0 = time.time()
            l, a = run(cfg, s, mode)
            ref = z[f'losses_{s}'][:cfg['steps']] if f'losses_{s}' in z.files else l
            rel = np.abs(l - ref) / np.maximum(np.abs(ref), 1e-6)
            accs.append(a.mean()); tails.append(np.median(l[-T:]))
            print(f'  {mode} seed {s}: held-out accuracy {a.mean():.3f} ({" ".join(f"{v:.3f}" for v in a)}) tail loss median {np.median(l[-T:]):.4f}; vs reference: steps 0-19 {rel[:20].max():.1e}, 0-99 {rel[:100].max():.1e} [{time.time() - t0:.0f} s]', flush=True)
        print(f'{mode}: accuracy {min(accs):.3f} .. {max(accs):.3f} (mean {np.mean(accs):.4f} +- {np.std(accs, ddof=1) / np.sqrt(len(accs)):.4f}, std {np.std(accs, ddof=1):.3f}, n {len(accs)}); tail median geo-mean {np.exp(np.mean(np.log(tails))):.4f}', flush=True)


if __name__ == '__main__':
    main()
