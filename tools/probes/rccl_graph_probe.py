"""Can a torch.distributed (RCCL) all_reduce be captured into a HIP graph on this stack?  1-rank probe (the only kind a 1-GPU box
allows): python tools/probes/rccl_graph_probe.py"""
import os

import torch
import torch.distributed as dist

os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
t = torch.ones(1 << 20, device='cuda')
dist.all_reduce(t)                                   # communicator creation outside the capture
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        t.mul_(2.0)
        w = dist.all_reduce(t, async_op=True)
        w.wait()
        t.add_(1.0)
    t.fill_(1.0)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print('captured + replayed: t[0] =', float(t[0]), '(expected 15.0)')
except Exception as e:                               # noqa: BLE001
    print('capture failed:', type(e).__name__, str(e)[:300])
dist.destroy_process_group()
