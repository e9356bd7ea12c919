#!/usr/bin/env python
"""Where does the segmented (multi-GPU) step lose time against the single-graph step?  1 GPU, 1-rank RCCL group (probe)."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simple3d_former_amd as s3d  # noqa: E402
from simple3d_former_amd.parallel import DataParallelTrainer  # noqa: E402
from oracle import voxel_oracle as vo  # noqa: E402  (synthetic inputs only)

os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29542')
dist.init_process_group('nccl', rank=0, world_size=1)
cfg = dict(backbone='deit_small_patch16_224', embed_layer='VoxelEmbed', voxel_size=32, cell=6, patch=5, n_classes=40)
x, y = vo.synthetic_batch(64, 32, 40, seed=9)
x, y = x.cuda(), y.cuda()


def run(nb, collectives, label):
    eng = s3d.VoxelEngine(**cfg)
    eng.load_state_dict(vo.init_state_dict(seed=9, **cfg))
    tr = DataParallelTrainer(eng, n_buckets=nb, force_collectives=True)
    if not collectives:
        tr.reducer.launch = lambda k: None
    cap = tr.capture(64); cap['x'].copy_(x); cap['y'].copy_(y)
    for _ in range(20):
        tr.step_graph()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        tr.step_graph()
    torch.cuda.synchronize()
    print(f'{label:45s} {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms/step', flush=True)


for nb in (1, 2, 4):
    run(nb, False, f'{nb} segment graph(s) + optimizer graph, no RCCL')
    run(nb, True, f'{nb} segment graph(s) + optimizer graph + RCCL')
