"""Same-process A/B of a dispatch knob on the whole cfg-3 training step (batch 64, dropout 0.1, HIP-graph replay): alternates s3d_debug_knob settings,
re-capturing the step graph for each, and prints ms per step (tools only).   python tools/r6/cfg3_knob_ab.py <knob id> <value a> <value b> [rounds=3] [steps=5]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import simple3d_former_amd as s3d
from simple3d_former_amd import _lib as L
from oracle import voxel_oracle as vo
kid, va, vb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
kw = dict(backbone='deit_base_patch16_224', embed_layer='VoxelEmbed_no_average', voxel_size=128, cell=9, patch=14, n_classes=55)
sd = vo.init_state_dict(seed=9, pos_embedding='group_embed', **kw)
x, y = vo.synthetic_batch(64, 128, 55, seed=9)
x, y = x.cuda(), y.cuda()
lib = L.lib()
for r in range(rounds):
    for v in (va, vb):
        lib.s3d_debug_knob(kid, v)
        eng = s3d.VoxelEngine(device='cuda', pos_embedding='group_embed', **kw); eng.load_state_dict(sd); eng.set_optimizer(lr=1e-3); eng.set_dropout(0.1, seed=9)
        for _ in range(2): eng.train_step(x, y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): eng.train_step(x, y)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / steps * 1e3
        print(f'round {r}: knob {kid} = {v:2d}: {ms:8.2f} ms per step (eager launches)', flush=True)
        del eng; torch.cuda.empty_cache()
