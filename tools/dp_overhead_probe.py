#!/usr/bin/env python
"""What the launch structure of the data-parallel step costs on ONE GPU (1-rank RCCL group, collectives forced): single graph without
collectives vs one graph per backward segment vs ONE graph with event-record nodes (collectives on a side stream) vs collectives
captured in the graph; and the event variants with the collectives suppressed (what the events / stream hops alone cost).

    python tools/dp_overhead_probe.py [n_buckets]"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simple3d_former_amd as s3d  # noqa: E402
from simple3d_former_amd.parallel import DataParallelTrainer  # noqa: E402
from oracle import voxel_oracle as vo  # noqa: E402

CFG = dict(backbone='deit_small_patch16_224', embed_layer='VoxelEmbed', voxel_size=32, cell=6, patch=5, n_classes=40)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 4
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
dist.init_process_group('nccl', rank=0, world_size=1)
sd = vo.init_state_dict(seed=9, **CFG)
x, y = vo.synthetic_batch(64, 32, 40, seed=9)
x, y = x.cuda(), y.cuda()


def run(name, skip=False, **kw):
    eng = s3d.VoxelEngine(device='cuda', **CFG)
    eng.load_state_dict(sd)
    tr = DataParallelTrainer(eng, n_buckets=NB, use_graphs=True, **kw)
    cap = tr.capture(64)
    cap['x'].copy_(x); cap['y'].copy_(y)
    tr.reducer.skip = skip
    for _ in range(20):
        tr.step_graph()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(200):
            tr.step_graph()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 200 * 1e3)
    print(f'{name:64s} {best:.4f} ms/step   ({tr.collectives_mode()}, {len(tr.slices)} buckets)')
    return best


base = run('single graph, no collectives', force_collectives=False)
for name, kw, skip in [('one graph per segment + host-launched all-reduces', dict(force_collectives=True, event_graph=False), False),
                       ('one graph per segment, collectives suppressed', dict(force_collectives=True, event_graph=False), True),
                       ('ONE graph + event nodes, all-reduces from a side stream', dict(force_collectives=True), False),
                       ('ONE graph + event nodes, collectives suppressed', dict(force_collectives=True), True),
                       ('all-reduces captured inside ONE graph', dict(force_collectives=True, graph_collectives=True), False),
                       ('bf16 wire: ONE graph + event nodes', dict(force_collectives=True, wire='bf16'), False),
                       ('bf16 wire: one graph per segment', dict(force_collectives=True, wire='bf16', event_graph=False), False)]:
    t = run(name, skip=skip, **kw)
    print(f'{"":64s} {100 * (t / base - 1):+.1f} % vs the single graph')
dist.destroy_process_group()
