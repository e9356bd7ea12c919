import sys, torch
sys.path.insert(0, '/root/repo')
from simple3d_former_amd import ops
Bb, H, N, hd = 4, 6, 26, 64
D = H * hd
qkv = torch.randn(Bb * N, 3 * D, device='cuda')
hi, lo = ops.split_bf16(qkv)
out_hi, out_lo, lse = ops.attention_fwd(hi, lo, Bb, H, N, D, N, 1, split=True)
torch.cuda.synchronize(); print('fwd ok', flush=True)
dout = torch.randn(Bb * N, D, device='cuda').to(torch.bfloat16)
d = ops.attention_bwd(hi, out_hi, out_lo, lse, dout, Bb, H, N, D, N, 1)
torch.cuda.synchronize(); print('bwd ok', float(d.float().abs().sum()), flush=True)
