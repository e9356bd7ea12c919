"""Numerics probe: the seq-first encoder attention with the P V product on ONE bf16 plane of P (P_hi . (V_hi + V_lo): two MFMAs
instead of three) -- max |d logits| against the fp32 oracle, cfg-3 real geometry, for a diffuse (reference initialisation) and for sharpened
attention (q / k projection scaled up), with the softmax denominator from the exact or from the rounded probabilities.  CPU only."""
import sys, os
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import voxel_oracle as vo

def rb(t): return t.to(torch.bfloat16).to(torch.float32)
MODE = {'v': 'exact'}

def enc(x, sd, dropout_p=0.0, training=False, generator=None, hash_seed=None):
    g = 'group_embed.'
    L, Nb, D = x.shape
    H, hd = vo.GROUP_HEADS, D // vo.GROUP_HEADS
    qkv = x @ sd[g + 'self_attn.in_proj_weight'].t() + sd[g + 'self_attn.in_proj_bias']
    q, k, v = qkv.split(D, dim=-1)
    q = q.reshape(L, Nb, H, hd).permute(1, 2, 0, 3) * hd ** -0.5
    k = k.reshape(L, Nb, H, hd).permute(1, 2, 0, 3)
    v = v.reshape(L, Nb, H, hd).permute(1, 2, 0, 3)
    s = q @ k.transpose(-2, -1)
    pu = torch.exp(s - s.max(-1, keepdim=True).values)                 # the kernel's unnormalised probabilities in (0, 1]
    if MODE['v'] == 'exact':
        a = (pu @ v) / pu.sum(-1, keepdim=True)
    elif MODE['v'] == 'round_num':                                     # numerator on bf16(P), denominator exact
        a = (rb(pu) @ v) / pu.sum(-1, keepdim=True)
    else:                                                              # both on bf16(P)
        a = (rb(pu) @ v) / rb(pu).sum(-1, keepdim=True)
    MODE['ent'] = float((-(pu / pu.sum(-1, keepdim=True)) * torch.log((pu / pu.sum(-1, keepdim=True)).clamp_min(1e-30))).sum(-1).mean())
    a = a.permute(2, 0, 1, 3).reshape(L, Nb, D)
    a = a @ sd[g + 'self_attn.out_proj.weight'].t() + sd[g + 'self_attn.out_proj.bias']
    x = F.layer_norm(x + a, (D,), sd[g + 'norm1.weight'], sd[g + 'norm1.bias'], vo.GROUP_LN_EPS)
    f = F.relu(x @ sd[g + 'linear1.weight'].t() + sd[g + 'linear1.bias'])
    f = f @ sd[g + 'linear2.weight'].t() + sd[g + 'linear2.bias']
    return F.layer_norm(x + f, (D,), sd[g + 'norm2.weight'], sd[g + 'norm2.bias'], vo.GROUP_LN_EPS)

def main():
    torch.set_num_threads(int(os.environ.get('THREADS', '8')))
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    kw = dict(backbone='deit_base_patch16_224', embed_layer='VoxelEmbed_no_average', voxel_size=128, cell=9, patch=14, n_classes=55)
    fk = dict(backbone=kw['backbone'], embed_layer=kw['embed_layer'], cell=9, patch=14, pos_embedding='group_embed')
    vo.group_encoder_layer = enc
    for seed in (9, 10):
        for sharp in (1.0, 8.0, 30.0):
            sd = vo.init_state_dict(seed=seed, pos_embedding='group_embed', exercise_all=True, **kw)
            D = 768
            w = sd['group_embed.self_attn.in_proj_weight']
            w[:2 * D] *= sharp                                         # q and k projections: scores x sharp^2
            x, _ = vo.synthetic_batch(B, 128, 55, seed=seed)
            out = {}
            with torch.no_grad():
                for m in ('exact', 'round_num', 'round_both'):
                    MODE['v'] = m
                    out[m] = vo.forward(sd, x, **fk)
            print(f'seed {seed} q/k x{sharp:g}: attention entropy {MODE["ent"]:.2f} nats (ln N = {torch.log(torch.tensor(196.0 * B)):.2f}); max |d logits| numerator-only '
                  f'{float((out["round_num"] - out["exact"]).abs().max()):.2e}, numerator + denominator {float((out["round_both"] - out["exact"]).abs().max()):.2e}; '
                  f'|logits| max {float(out["exact"].abs().max()):.2f}', flush=True)

if __name__ == '__main__':
    main()
