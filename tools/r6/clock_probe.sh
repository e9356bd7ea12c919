#!/bin/bash
# shader clock / power while the seq-first attention forward runs back to back (is the kernel clock- / power-bound?)
for knob in "" "0:3" "0:2"; do
  ( timeout 120 python - "$knob" <<'PY'
import sys, os, torch, ctypes
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from simple3d_former_amd import _lib as L, ops
lib = L.lib()
if sys.argv[1]:
    k, v = sys.argv[1].split(':'); lib.s3d_debug_knob(int(k), int(v))
Bb, H, hd, B = 15, 4, 192, 64
N, D = B * 196, H * hd
g = torch.Generator().manual_seed(6)
qkv = (torch.randn(Bb * N, 3 * D, generator=g) * 0.5).cuda()
hi, lo = ops.split_bf16(qkv); del qkv
seed = torch.tensor([4321], dtype=torch.int64, device='cuda')
T = (N + 31) // 32
mbuf = torch.zeros(Bb * H * T * T * 32 + 256, dtype=torch.int32, device='cuda')
import time
t0 = time.time(); n = 0
while time.time() - t0 < 8:
    ops.attention_fwd(hi, lo, Bb, H, N, D, 1, Bb, split=True, drop=(0.1, seed, 0), drop_mask=mbuf); n += 1
    if n % 20 == 0: torch.cuda.synchronize()
torch.cuda.synchronize()
print(f'knob {sys.argv[1] or "shipped"}: {n} launches in {time.time() - t0:.2f} s = {(time.time() - t0) / n * 1e3:.2f} ms each')
PY
  ) &
  sleep 5
  for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket Graphics Package Power" | tr '\n' ' '; echo; sleep 0.7; done
  wait
done
echo idle:; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
