"""Child process of parallel.graph_collectives_preflight: proves, with the REAL peers of the job, that RCCL all-reduces captured in
a HIP graph replay correctly -- before the trainer bakes the gradient all-reduces of DDP (train_cls_voxel.py:155-159, :287) into its step
graph.  A capture problem that only shows with peers hangs inside the device queue and cannot be recovered in-process, so the trial
runs in a throw-away process group (same ranks and devices, its own rendezvous port): the parent kills it on a timeout and falls back
to host-launched collectives between graph segments.  Two buckets are put in flight side by side, as the step graph does."""
import datetime
import os
import sys


def main():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    if os.environ.get('S3D_PREFLIGHT_FORCE_FAIL') == '1':           # tests: exercise the parent's fallback path
        print('S3D_PREFLIGHT_FORCED_FAILURE', flush=True)
        sys.exit(5)
    torch.cuda.set_device(int(os.environ['S3D_PREFLIGHT_DEVICE']))
    dist.init_process_group(os.environ.get('S3D_PREFLIGHT_BACKEND', 'nccl'), init_method='env://', world_size=world, rank=rank,
                            timeout=datetime.timedelta(seconds=float(os.environ.get('S3D_PREFLIGHT_INIT_TIMEOUT', '60'))))
    n = 1 << 20                                                   # 4 MB per bucket: past RCCL's small-message protocols
    buf = torch.zeros(2, n, device='cuda')
    dist.all_reduce(buf[0])                                       # communicator set-up outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        buf[0].fill_(float(rank + 1))
        w0 = dist.all_reduce(buf[0], async_op=True)
        buf[1].fill_(float(2 * (rank + 1)))
        w1 = dist.all_reduce(buf[1], async_op=True)
        w0.wait(); w1.wait()
        buf.mul_(0.5)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    want = world * (world + 1) / 2
    ok = bool((buf[0] == 0.5 * want).all()) and bool((buf[1] == want).all())
    if ok and os.environ.get('S3D_PREFLIGHT_SHARDED') == '1':
        # the sharded trainer's pair: IN-PLACE reduce-scatter of a bucket (output = the rank's chunk of the input) and in-place all-gather
        # (input = the rank's chunk of the output) on a side stream of a captured graph, as parallel.ShardedDataParallelTrainer issues them
        m = (1 << 20) // world * world
        b2 = torch.zeros(m, device='cuda')
        c = m // world
        dist.reduce_scatter_tensor(b2[rank * c:(rank + 1) * c], b2)
        dist.all_gather_into_tensor(b2, b2[rank * c:(rank + 1) * c])
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            b2.fill_(float(rank + 1))
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                dist.reduce_scatter_tensor(b2[rank * c:(rank + 1) * c], b2)
                b2[rank * c:(rank + 1) * c].add_(float(rank))          # "the shard's update"
                dist.all_gather_into_tensor(b2, b2[rank * c:(rank + 1) * c])
            torch.cuda.current_stream().wait_stream(side)
        for _ in range(3):
            g2.replay()
        torch.cuda.synchronize()
        expect = torch.cat([torch.full((c,), want + r, device='cuda') for r in range(world)])
        ok = bool((b2 == expect).all())
        if not ok:
            print(f'S3D_PREFLIGHT_WRONG sharded pair: {b2[::c].tolist()} want {expect[::c].tolist()}', flush=True)
    dist.destroy_process_group()
    print('S3D_PREFLIGHT_OK' if ok else f'S3D_PREFLIGHT_WRONG {float(buf[0, 0])} {float(buf[1, 0])} want {0.5 * want} {want}', flush=True)
    sys.exit(0 if ok else 4)


if __name__ == '__main__':
    main()
