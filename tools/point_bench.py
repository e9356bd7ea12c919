#!/usr/bin/env python
"""Throughput of the point-cloud training step (forward + CE + backward + SGD) on one MI355X.
   python tools/point_bench.py cfg4|cfg5 [steps]     cfg4: cls 1024 pts x 6, B=128; cfg5: seg 2048 pts x 22, B=32
   VARIANT=3DViT_1_layer|3DViT_0_layer|3DViT_LWF BACKBONE=deit_small_patch16_224 select the other part-seg model directories
   (config/model/3DViT_*.yaml) on the cfg5 data shape; LWF=1 adds the image branch of train_partseg_lwf.py (IMG_BATCH images).
   Data parallel (cfg-4 on 4 GPUs, cfg-5 on 8; weak scaling, `batch` clouds per GPU), one process per GPU over RCCL:
   python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/point_bench.py cfg5
   (rank 0 prints the whole-job rate; FORCE_COLLECTIVES=1 issues the all-reduces at N = 1 too).
   PIPELINE=1: train_step_pipelined -- the geometry (FPS / kNN) of batch i+1 is computed on the side stream during step i."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple3d_former_amd.point_engine import PointEngine  # noqa: E402
from oracle import point_oracle as po  # noqa: E402  (synthetic-input recipe only)

CFG = {'cfg4': dict(task='cls', n_points=1024, d_points=6, n_classes=40, batch=128),
       'cfg5': dict(task='seg', n_points=2048, d_points=22, n_classes=50, batch=32)}


def main():
    import torch.distributed as dist
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (('RANK', 0), ('WORLD_SIZE', 1), ('LOCAL_RANK', 0)))
    force = os.environ.get('FORCE_COLLECTIVES', '0') == '1'
    torch.cuda.set_device(local)
    if world > 1 or force:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world)
    name = sys.argv[1] if len(sys.argv) > 1 else 'cfg4'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    c = CFG[name]
    B = int(os.environ.get('BATCH', c['batch']))
    variant = os.environ.get('VARIANT', '3DViT')
    backbone = os.environ.get('BACKBONE', 'deit_tiny_patch16_224')
    lwf = os.environ.get('LWF', '0') == '1'
    eng = PointEngine(backbone=backbone, n_points=c['n_points'], d_points=c['d_points'], n_classes=c['n_classes'],
                      task=c['task'], device='cuda', variant=variant)
    eng.load_state_dict(po.init_state_dict(backbone=backbone, n_classes=c['n_classes'], d_points=c['d_points'], seed=9, variant=variant))
    x, y, starts = po.synthetic_points(B, c['n_points'], c['d_points'], c['n_classes'], c['task'], seed=9 + rank, variant=variant)
    x, y, starts = x.cuda(), y.cuda(), tuple(s.cuda() for s in starts)
    if lwf:
        Bi = int(os.environ.get('IMG_BATCH', B))
        img = (torch.rand(Bi, 3, 224, 224, generator=torch.Generator().manual_seed(9)) * 2 - 1).cuda()
        yi = torch.randint(0, 1000, (Bi,), generator=torch.Generator().manual_seed(10)).cuda()
        train = lambda: eng.lwf_train_step(x, y, starts, img, yi, 0.1)[0]
    else:
        train = lambda: eng.train_step(x, y, starts)
    for _ in range(3):
        loss = train()
    torch.cuda.synchronize()
    use_graph = os.environ.get('GRAPH', '1') != '0' and not lwf
    dp = dist.is_initialized() and not lwf
    if dp:                                          # graphs [fwd, CE, bwd top] | RCCL | [bwd bottom] | RCCL | [SGD]
        from simple3d_former_amd.parallel import PointDataParallelTrainer
        tr = PointDataParallelTrainer(eng, use_graphs=use_graph, force_collectives=force)
        loss = tr.step(x, y, starts)
        step = (lambda: tr.step_graph()) if use_graph else (lambda: tr.step_eager(x, y, starts))
    elif os.environ.get('PIPELINE', '0') == '1' and not lwf:
        xs, ys, sts = [x, x.clone()], [y, y.clone()], [starts, tuple(t.clone() for t in starts)]
        state = {'p': 0}
        if use_graph:
            graphs, loss = eng.capture_train_step_pipelined(xs, ys, sts)

            def step():
                graphs[state['p']].replay()
                state['p'] ^= 1
        else:
            eng.prepare_geometry(xs[0], sts[0], 0)

            def step():
                p = state['p']
                eng.train_step_pipelined(xs[p], ys[p], sts[p], xs[1 - p], sts[1 - p], p)
                state['p'] ^= 1
    elif use_graph:                                   # the step has no host synchronisation: capture it once, replay it
        graph, loss = eng.capture_train_step(x, y, starts)
        step = graph.replay
    else:
        step = train
    step()
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    el = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([el], device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t)
    B *= world                                      # whole-job figures
    if rank != 0:
        dist.destroy_process_group()
        return
    out = dict(config=name, variant=variant, backbone=backbone, lwf=lwf, n_gpus=world, scaling='weak', global_batch=B, ms_per_step=round(el / steps * 1e3, 3), clouds_per_sec=round(B * steps / el, 1),
               points_per_sec=round(B * c['n_points'] * steps / el, 0), loss=round(float(loss), 5), launch=('hipGraph replay' if use_graph else 'eager') + (', geometry one step ahead' if os.environ.get('PIPELINE', '0') == '1' and not lwf and not dp else ''))
    import ctypes
    ctypes.CDLL(None).fflush(None)                  # RCCL's version banner sits in the C stdout buffer: out before the JSON line
    print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
