#!/usr/bin/env python
"""Headline benchmark: voxel-grids/s of the full training step (zero_grad -> forward -> cross-entropy -> backward ->
gradient all-reduce -> Adam) on BASELINE.json configs[1]: deit_small_patch16_224 + VoxelEmbed (32^3 grid, cell 6,
patch 5, 40 classes), batch 64 per GPU, synthetic 10 %-occupancy grids, random-init weights (reference init).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W)

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      dominant kernel (the MFMA GEMM instantiation with the largest total time), algorithmic 2*M*N*K flops per
                launch / its average launch duration measured with HIP events on the launch stream in an instrumented
                eager pass of the same steps, against the 2.5 PFLOP/s dense bf16 MFMA peak
  cpu_baseline  the CPU oracle's full training step (PyTorch fp32 restatement of the reference, same batch) timed on the
                host cores of this box, rank 0 at N=1 only, bounded to ~10-20 s
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE.json configs[1] -- the configuration the headline metric is quoted on (default)
    'cfg2': dict(cfg=dict(backbone='deit_small_patch16_224', embed_layer='VoxelEmbed', voxel_size=32, cell=6, patch=5,
                          n_classes=40), pos_embedding='default', batch=64, train_flops=3.39e9,
                 metric='voxels/sec (train, whole node) deit_small VoxelEmbed 32^3 b64',
                 workload='BASELINE.json configs[1]: deit_small_patch16_224 + VoxelEmbed(voxel 32, cell 6, patch 5), 40 classes, '
                          'full train step incl. Adam'),
    # BASELINE.json configs[2] (secondary; eval-mode dropout in the group encoder layer)
    'cfg3': dict(cfg=dict(backbone='deit_base_patch16_224', embed_layer='VoxelEmbed_no_average', voxel_size=128, cell=9,
                          patch=14, n_classes=55), pos_embedding='group_embed', batch=64, train_flops=2.0e12,
                 metric='voxels/sec (train, whole node) deit_base(H=3) VoxelEmbed_no_average 128^3 group_embed b64',
                 workload='BASELINE.json configs[2]: deit_base_patch16_224 (3 heads) + VoxelEmbed_no_average(voxel 128, cell 9, '
                          'patch 14) + group_embed, 55 classes, full train step incl. Adam'),
}
CFG = CONFIGS['cfg2']['cfg']
BATCH_PER_GPU = 64
TRAIN_FLOPS_PER_SAMPLE = 3.39e9          # BASELINE.md section 2 (fwd 1.137 G, train = 3x fwd - tokenizer dgrad)
MFMA_BF16_PEAK_TFLOPS = 2500.0           # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16
EPI_NAMES = {0: 'bf16_bias', 1: 'gelu', 2: 'resid', 3: 'token', 4: 'f32', 5: 'dgelu', 6: 'atomic', 7: 'relu', 8: 'drelu'}


def _key_fields(key):
    key = int(key)
    kind, rest = key // 10 ** 11, key % 10 ** 11
    return kind, rest // 10 ** 8, (rest // 10 ** 5) % 1000, rest % 10 ** 5


def kernel_name(key):
    """Decodes the library's profiling key (gemm.hip).  First digit = kernel family: 1 register-staged gemm_kernel
    (BM|BN|TA|TB|SPLIT|EPI), 2 register-staged dgrad+wgrad pair (BMdgrad|BMwgrad|EPI), 3 LDS-DMA forward kernel, 4 LDS-DMA kernel
    with transpose reads for k-major operands, 6 the dgrad+wgrad pair on that pipeline."""
    kind, b0, b1, tail = _key_fields(key)
    epi = EPI_NAMES.get(tail % 100, tail % 100)
    ta, tb, sp = (tail // 10000) % 10, (tail // 1000) % 10, (tail // 100) % 10
    lay = f'{"T" if ta else "N"}{"T" if not tb else "N"}'
    if kind == 2:
        return f'gemm_pair_kernel<dgrad {b0}x64 NN {epi} || wgrad {b1}x64 TN atomic>'
    if kind == 6:
        return f'gemm_pair_dmat_kernel<dgrad {b0}x{b1} NN {epi} || wgrad {b0}x{b1} TN atomic; LDS-DMA + transpose reads>'
    if kind == 3:
        return f'gemm_nt_dma_kernel<{b0},{b1},NT,{"split3" if sp else "bf16"},{epi}>'
    if kind == 4:
        return f'gemm_dmat_kernel<{b0},{b1},{lay},bf16,{epi}>'
    return f'gemm_kernel<{b0},{b1},{lay},{"split3" if sp else "bf16"},{epi}>'


def rocprof_name(key):
    """The same kernel as rocprofv3 prints it (template arguments), for looking it up in profiles/*.json."""
    kind, b0, b1, tail = _key_fields(key)
    tf = lambda v: 'true' if v else 'false'
    ta, tb, sp, epi = (tail // 10000) % 10, (tail // 1000) % 10, (tail // 100) % 10, tail % 100
    if kind == 2:
        return f'gemm_pair_kernel<{b0}, {epi}, {b1}>'
    if kind == 6:
        return f'gemm_pair_dmat_kernel<{epi}, 3>'
    if kind == 3:
        return f'gemm_nt_dma_kernel<{tf(sp)}, {epi}, 2, {32 if (sp and b0 == 128) else 64}, {b0}, {b1}>'
    if kind == 4:
        return f'gemm_dmat_kernel<{tf(ta)}, {tf(tb)}, {epi}, {2 if b0 == 128 else 3}, {b0}, {b1}>'
    return f'gemm_kernel<{b0}, {b1}, {tf(ta)}, {tf(tb)}, {tf(sp)}, {epi}>'


def pmc_traffic(key):
    """HBM-side bytes per launch of this kernel from the committed PMC passes (tools/pmc_step.sh -> profiles/), or None.
    bench.py cannot collect PMC counters itself (they need rocprofv3 around the process, one pass per counter group)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_pmc_step_traffic.json')
    try:
        with open(path) as f:
            k = json.load(f)['kernels'].get(rocprof_name(key))
        return (k['hbm_bytes_per_launch'], k) if k else (None, None)
    except (OSError, ValueError, KeyError):
        return None, None


def cpu_baseline(x, y, budget_s=15.0):
    """The oracle's training step (forward + CE + autograd backward + Adam on every used parameter) on host cores."""
    from oracle import voxel_oracle as vo
    sd = vo.init_state_dict(seed=9, voxel_size=CFG['voxel_size'], **{k: CFG[k] for k in ('backbone', 'embed_layer', 'cell', 'patch', 'n_classes')})
    names = vo.used_param_names(sd)
    m = {k: torch.zeros_like(sd[k]) for k in names}
    v = {k: torch.zeros_like(sd[k]) for k in names}
    kw = dict(backbone=CFG['backbone'], embed_layer=CFG['embed_layer'], cell=CFG['cell'], patch=CFG['patch'])
    threads = torch.get_num_threads()

    def one(step):
        _, loss, grads = vo.loss_and_grads(sd, x, y, **kw)
        for k, g in grads.items():
            vo.adam_step(sd[k], g, m[k], v[k], step)
        return float(loss)

    one(1)                                                     # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        one(n + 2)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 50:
            break
    return dict(value=round(n * x.shape[0] / el, 2), unit='voxels/sec', cores=threads, kind='port',
                sample=f'{n} full training steps (fwd+bwd+Adam) of the fp32 PyTorch-CPU oracle at batch {x.shape[0]}, '
                       f'{threads} threads, {el:.1f} s')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-graphs', action='store_true', help='eager launches instead of HIP-graph replay')
    ap.add_argument('--plain-bf16', action='store_true', help='one-MFMA forward (fails the 1e-3 logit bar; for comparison only)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--force-collectives', action='store_true',
                    help='diagnostic: run the segmented multi-GPU step (RCCL calls between graph segments) even at N=1')
    ap.add_argument('--graph-collectives', action='store_true',
                    help='experimental: one HIP graph per step with the RCCL all-reduces captured inside (default: one graph per '
                         'backward segment, collectives launched from the host in between)')
    ap.add_argument('--buckets', type=int, default=4)
    ap.add_argument('--config', choices=sorted(CONFIGS), default='cfg2')
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch override (non-headline experiments)')
    args = ap.parse_args()
    global CFG, BATCH_PER_GPU, TRAIN_FLOPS_PER_SAMPLE
    conf = CONFIGS[args.config]
    CFG, BATCH_PER_GPU, TRAIN_FLOPS_PER_SAMPLE = conf['cfg'], args.batch or conf['batch'], conf['train_flops']
    if args.config != 'cfg2':
        args.no_cpu_baseline = True

    import torch.distributed as dist
    import simple3d_former_amd as s3d
    from simple3d_former_amd import _lib as L
    from simple3d_former_amd.parallel import DataParallelTrainer
    from oracle import voxel_oracle as vo                      # synthetic-input recipe + cpu_baseline leg only

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'
    # S3D_BENCH_DEVICE / S3D_BENCH_BACKEND: dry run of the multi-rank code path on a box with fewer GPUs than ranks (all ranks on
    # one device, gloo collectives -- RCCL refuses two ranks per device); the numbers of such a run mean nothing
    dev_index = int(os.environ.get('S3D_BENCH_DEVICE', local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1 or args.force_collectives:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29541')
        dist.init_process_group(backend=os.environ.get('S3D_BENCH_BACKEND', 'nccl'), init_method='env://', world_size=world, rank=rank)

    # model + optimizer state (reference init, seed 9), per-rank synthetic shard of the global batch
    eng = s3d.VoxelEngine(device=dev, split=not args.plain_bf16, pos_embedding=conf['pos_embedding'], **CFG)
    sd = vo.init_state_dict(seed=9, pos_embedding=conf['pos_embedding'], **CFG)
    eng.load_state_dict(sd)
    x_cpu, y_cpu = vo.synthetic_batch(BATCH_PER_GPU, CFG['voxel_size'], CFG['n_classes'], seed=9 + rank)
    x, y = x_cpu.to(dev), y_cpu.to(dev)
    trainer = DataParallelTrainer(eng, n_buckets=args.buckets, use_graphs=not args.no_graphs,
                                  force_collectives=args.force_collectives, graph_collectives=args.graph_collectives)
    trainer.set_optimizer(lr=1e-3)                              # README recipe (README.md:60)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if not args.no_graphs:
        cap = trainer.capture(BATCH_PER_GPU)
        cap['x'].copy_(x); cap['y'].copy_(y)
        step = trainer.step_graph
    else:
        step = lambda: trainer.step_eager(x, y)

    first_loss = None
    for i in range(args.warmup):
        l = step()
        if i == 0:
            first_loss = float(l)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    final_loss = float(loss)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms = elapsed / args.steps * 1e3
    value = world * BATCH_PER_GPU * args.steps / elapsed

    out = {
        'metric': conf['metric'], 'value': round(value, 1),
        'unit': 'voxels/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 4),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'bf16',
        'precision_note': ('plain bf16 forward (fails the 1e-3 logit bar; comparison only)' if args.plain_bf16 else
                           'bf16 MFMA operands: split-bf16 (hi+lo, 3 MFMAs per product) forward, plain bf16 backward; fp32 '
                           'accumulation, residual stream, LayerNorm/softmax/GELU, loss and Adam'),
        'data': f'synthetic (seeded 10%-occupancy {CFG["voxel_size"]}^3 grids, random-init weights of the reference architecture)',
        'config': {'workload': conf['workload'], 'batch_per_gpu': BATCH_PER_GPU,
                   'global_batch': world * BATCH_PER_GPU, 'tokens_per_sample': eng.ntok, 'parallelism': f'dp{world}',
                   'launch': 'eager' if args.no_graphs else 'hipGraph replay',
                   'grad_buckets': len(trainer.slices),
                   'collectives': ('none' if not (world > 1 or args.force_collectives) else 'captured in the step graph'
                                   if args.graph_collectives else 'host-launched between graph segments')},
        'voxel_cells_per_sec': round(value * CFG['voxel_size'] ** 3, 0),
        'algorithmic_tflops': round(value * TRAIN_FLOPS_PER_SAMPLE / 1e12, 2),
        'loss_first_step': round(first_loss, 5) if first_loss is not None else None, 'loss_last_step': round(final_loss, 5),
    }

    if rank == 0 and not args.no_roofline:
        # instrumented eager pass: HIP events around every GEMM launch, on the launch stream
        lib = L.lib()
        n_inst = max(3, min(10, args.steps))
        lib.s3d_prof_enable(1)
        for _ in range(n_inst):
            eng.train_step(x, y)          # engine-only step (no collective): rank-0 kernel timing
        torch.cuda.synchronize()
        import ctypes
        rows = (ctypes.c_double * (4 * 64))()
        n = lib.s3d_prof_collect(rows, 64)
        lib.s3d_prof_enable(0)
        ov = ctypes.c_double(0.0)                                           # what the event bracket itself adds per launch
        lib.s3d_prof_event_overhead(L.current_stream(), ctypes.byref(ov))
        ov_ms = ov.value * 1e-3
        raw = [(rows[4 * i], rows[4 * i + 1], rows[4 * i + 2], rows[4 * i + 3]) for i in range(min(n, 64))]
        ev = [(k, cnt, max(ms_ - cnt * ov_ms, 1e-9), fl) for k, cnt, ms_, fl in raw]      # event-bracket timings
        ks, method = ev, ('HIP events on the launch stream around every GEMM launch (instrumented eager pass), minus the time an '
                          'empty event pair measures')
        if ev and not args.no_graphs and not eng.group:
            # Difference timing inside the busy graph: the step is captured once more with one GEMM instantiation suppressed
            # (s3d_prof_skip); (t_full - t_without) / launches is that kernel's duration under the timed region's own conditions
            # (back-to-back kernels, sustained clocks).  The eager pass above is CPU-paced: the GPU idles between launches and
            # the same kernels read ~10 % faster there than rocprofv3 shows for the graph replay.
            def graph_ms(reps=60):
                eng._graphs.clear()
                g, sx, sy, _ = eng.capture_train_step(BATCH_PER_GPU)
                sx.copy_(x); sy.copy_(y)
                for _ in range(5):
                    g.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / reps
            t_full = graph_ms()
            diff = []
            for k, cnt, ms_, fl in ev:
                lib.s3d_prof_skip(ctypes.c_double(k))
                t_wo = graph_ms()
                lib.s3d_prof_skip(ctypes.c_double(0.0))
                per_step = cnt / n_inst
                diff.append((k, cnt, max(t_full - t_wo, 1e-9) * n_inst, fl))           # same units as ev: ms over n_inst steps
            eng._graphs.clear()
            ks, method = diff, ('difference timing in the replayed HIP graph: (step time - step time with this kernel suppressed) / '
                                'launches per step, HIP events around 60 replays each')
        if ks:
            tot_ms = sum(k[2] for k in ks)
            dom = max(ks, key=lambda k: k[2])
            avg_us = dom[2] / dom[1] * 1e3
            achieved = dom[3] / (dom[2] * 1e-3) / 1e12                       # algorithmic TFLOP/s of the dominant kernel
            all_ach = sum(k[3] for k in ks) / (tot_ms * 1e-3) / 1e12
            traffic, tdetail = pmc_traffic(dom[0])
            ev_by_key = {k[0]: k for k in ev}
            out['roofline'] = {
                'bound': 'mfma', 'kernel': kernel_name(dom[0]), 'achieved': round(achieved, 2), 'peak': MFMA_BF16_PEAK_TFLOPS,
                'unit': 'TFLOP/s', 'frac': round(achieved / MFMA_BF16_PEAK_TFLOPS, 5), 'traffic': traffic,
                'traffic_unit': 'bytes per launch (memory side of L2: HBM + Infinity Cache)',
                'traffic_source': ('profiles/r01_pmc_step_traffic.json: rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + '
                                   'WRITE_SIZE, separate passes, ' + rocprof_name(dom[0])) if traffic else None,
                'traffic_detail': tdetail,
                'avg_launch_us': round(avg_us, 3), 'timing': method,
                'avg_launch_us_events': round(ev_by_key[dom[0]][2] / ev_by_key[dom[0]][1] * 1e3, 3),
                'event_bracket_overhead_us': round(ov.value, 3),
                'launches_per_step': round(dom[1] / n_inst, 1),
                'flops_per_launch': round(dom[3] / dom[1], 0),
                'mfma_issue_factor': 3 if 'split3' in kernel_name(dom[0]) else 1,
                'all_gemm_kernels': {'achieved': round(all_ach, 2), 'ms_per_step': round(tot_ms / n_inst, 4),
                                     'share_of_step': round(tot_ms / n_inst / ms, 3)},
                'per_kernel': [{'kernel': kernel_name(k[0]), 'launches_per_step': round(k[1] / n_inst, 1),
                                'avg_us': round(k[2] / k[1] * 1e3, 3), 'tflops': round(k[3] / (k[2] * 1e-3) / 1e12, 2),
                                'avg_us_events': round(ev_by_key[k[0]][2] / ev_by_key[k[0]][1] * 1e3, 3)}
                               for k in sorted(ks, key=lambda k: -k[2])],
            }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(x_cpu, y_cpu)

    # RCCL writes a version banner to the C stdout of every rank, block-buffered until the process exits -- i.e. AFTER a JSON line
    # printed from Python.  Push it out on every rank first, so that the JSON line is the last thing on stdout.
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    if world > 1:
        barrier()                     # rank 0 ran the instrumented pass; every rank's banner is out
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or args.force_collectives:
        if world > 1:
            barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
