"""Adam beside the backward WITHOUT a branch inside a graph: the backward is captured as `nb` chain graphs (one per gradient bucket,
as the data-parallel trainer does), and the Adam slice of bucket k is launched eagerly on a second stream behind an event recorded
after graph k -- two streams overlap as separate launches, no captured graph has two live branches (tools/probes/graph_branch_probe.py).
Compares with the one-graph step (update after the whole backward).  cfg-2, batch 64."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import simple3d_former_amd as s3d                           # noqa: E402
from oracle import voxel_oracle as vo                         # noqa: E402  (synthetic-input recipe only)

CFG = dict(backbone='deit_small_patch16_224', embed_layer='VoxelEmbed', voxel_size=32, cell=6, patch=5, n_classes=40)
dev = 'cuda'
B = 64
x, y = vo.synthetic_batch(B, 32, 40, seed=9)
x, y = x.to(dev), y.to(dev)


def engine():
    eng = s3d.VoxelEngine(device=dev, lr=1e-3, **CFG)
    eng.load_state_dict(vo.init_state_dict(seed=9, **CFG))
    return eng


def timeit(step, reps=300):
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


eng = engine()
g, sx, sy, loss = eng.capture_train_step(B)
sx.copy_(x); sy.copy_(y)
print(f'one graph, update behind the backward: {timeit(g.replay):.4f} ms')

for nb, cap in ((2, 0), (3, 0), (4, 0), (4, 512), (6, 0), (4, -1)):
    eng = engine()
    segs, slices = eng.grad_buckets(nb)
    side = torch.cuda.Stream()

    def phase(k):
        if k == 0:
            eng.advance_dropout_seed()
            eng.forward_loss(sx, sy)
            ws = eng.backward_begin(B)
        else:
            ws = eng.workspace(B)
        eng.backward_segment(ws, segs[k][0], segs[k][1], k == nb - 1)

    warm = torch.cuda.Stream()
    warm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(warm):
        for k in range(nb):
            phase(k)
        eng.adam_step()
    torch.cuda.current_stream().wait_stream(warm)
    torch.cuda.synchronize()
    graphs = []
    for k in range(nb):
        gk = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gk):
            phase(k)
        graphs.append(gk)
    evs = [torch.cuda.Event() for _ in range(nb)]
    serial = cap < 0

    def step():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            eng.adam_begin()
        for k in range(nb):
            graphs[k].replay()
            if serial:
                continue
            evs[k].record(main)
            side.wait_event(evs[k])
            with torch.cuda.stream(side):
                eng.adam_apply(*slices[k], max_workgroups=cap)
        main.wait_stream(side)
        if serial:
            eng.adam_apply(0, eng.arena.numel)

    what = 'update behind the last one (no overlap)' if serial else f'Adam slice k on a second stream behind graph k (grid cap {cap or 2048})'
    print(f'{nb} backward graphs, {what}: {timeit(step):.4f} ms')
