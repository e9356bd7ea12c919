#!/usr/bin/env python
"""Benchmarks of the full training step on synthetic inputs, one process per GPU.

    python bench.py --gpus N --steps K --warmup W [--config cfg2|cfg3|cfg4|cfg5]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W)

  cfg2 (default, the headline)  BASELINE.json configs[1]: deit_small_patch16_224 + VoxelEmbed (32^3 grid, cell 6, patch 5, 40 classes),
                                batch 64 per GPU: zero_grad -> forward -> cross-entropy -> backward -> gradient all-reduce -> Adam
  cfg3                          configs[2]: deit_base (3 heads) + VoxelEmbed_no_average 128^3 + group_embed (dropout 0.1), batch 64
  cfg4 / cfg5                   configs[3] / [4]: PointTransformerCls (1024 pts, batch 128) / PointTransformerSeg (2048 pts, batch 32):
                                FPS / kNN geometry -> forward -> CE -> backward -> all-reduce -> SGD(momentum); clouds/s (+ seg-pts/s)

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline      voxel configs: the dominant MFMA GEMM instantiation, algorithmic 2*M*N*K flops per launch / its average launch
                duration (difference timing inside the replayed HIP graph + HIP events on the launch stream), vs 2.5 PFLOP/s;
                point configs: the dominant HBM-bound operator (BatchNorm / neighbourhood gather), algorithmic bytes / its launch
                duration (HIP events on the launch stream) vs 8 TB/s, with the dominant GEMM nested under "mfma"
  cpu_baseline  the CPU oracle's full training step timed on this box's host cores, rank 0 at N = 1 only, bounded to ~15-30 s
                (cfg3 / cfg4 / cfg5: at a reduced batch, stated in "sample")
  N > 1         rccl_ranks / distinct_devices (the process group spans N GPUs), allreduce_ms[] per gradient bucket, the step time
                with the collectives suppressed and overlap_frac
"""
import argparse
import ctypes
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
    # BASELINE.json configs[2] (secondary; the group encoder layer trains with dropout 0.1 as the reference's
    # nn.TransformerEncoderLayer does under model.train(), vit_3d_2d_pretrain.py:381)
    'cfg3': dict(cfg=dict(backbone='deit_base_patch16_224', embed_layer='VoxelEmbed_no_average', voxel_size=128, cell=9,
                          patch=14, n_classes=55), pos_embedding='group_embed', batch=64, train_flops=2.0e12, dropout=0.1,
                 cpu_batch=1,
                 metric='voxels/sec (train, whole node) deit_base(H=3) VoxelEmbed_no_average 128^3 group_embed b64',
                 workload='BASELINE.json configs[2]: deit_base_patch16_224 (3 heads) + VoxelEmbed_no_average(voxel 128, cell 9, '
                          'patch 14) + group_embed (training-mode dropout 0.1), 55 classes, full train step incl. Adam'),
}
# BASELINE.json configs[3] / configs[4]: the point path (models/3DViT; train_cls.py:69-119, train_partseg.py:74-150)
POINT_CONFIGS = {
    'cfg4': dict(task='cls', n_points=1024, d_points=6, n_classes=40, batch=128, cpu_batch=8, fwd_flops=4.3e9,
                 metric='clouds/sec (train, whole node) PointTransformerCls deit_tiny 1024 pts b128',
                 workload='BASELINE.json configs[3]: ModelNet40 point-cloud cls, models/3DViT PointTransformerCls (deit_tiny), '
                          '1024 points x 6 channels, 40 classes, full train step incl. SGD(momentum)'),
    'cfg5': dict(task='seg', n_points=2048, d_points=22, n_classes=50, batch=32, cpu_batch=2, fwd_flops=9.8e9,
                 metric='clouds/sec (train, whole node) PointTransformerSeg deit_tiny 2048 pts b32 (+ seg-pts/sec)',
                 workload='BASELINE.json configs[4]: ShapeNetPart part-seg, models/3DViT PointTransformerSeg (deit_tiny), '
                          '2048 points x 22 channels (6 + one-hot 16), 50 parts, per-point head, full train step incl. SGD(momentum)'),
}
HBM_PEAK_GBPS = 8000.0                  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
CFG = CONFIGS['cfg2']['cfg']
BATCH_PER_GPU = 64
TRAIN_FLOPS_PER_SAMPLE = 3.39e9          # BASELINE.md section 2 (fwd 1.137 G, train = 3x fwd - tokenizer dgrad)
MFMA_BF16_PEAK_TFLOPS = 2500.0           # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16
EPI_NAMES = {0: 'bf16_bias', 1: 'gelu', 2: 'resid', 3: 'token', 4: 'f32', 5: 'dgelu', 6: 'atomic', 7: 'relu', 8: 'drelu'}


# round 5 (csrc/bwd_gemm.hip): key family -> (description, rocprofv3 name stem)
R5_KERNELS = {
    10: ('wgrad_group_kernel<128x128 TN, full k, grouped layers, plain stores> (bf16)', 'wgrad_group_kernel<4, 32'),
    11: ('dgrad_kernel<64x64 NN, k-slices as fp32 planes> (bf16)', 'dgrad_kernel<0, 3'),
    12: ('dgrad_kernel<64x64 NN, gelu\' + LayerNorm row statistics> (bf16)', 'dgrad_kernel<1, 3'),
    13: ('dgrad_kernel<64x64 NN, LayerNorm-backward epilogue, two wave groups> (bf16)', 'dgrad_kernel<2, 3'),
    14: ('dgrad_lnrows_kernel<64x192 NN, whole-row LayerNorm-backward epilogue> (bf16)', 'dgrad_lnrows_kernel<3'),
    15: ('blk_mlp_full_kernel<norm2 + fc1 + GELU + fc2 + residual, D = 192, one launch> (split bf16)', 'blk_mlp_full_kernel<16'),
}


def _key_fields(key):
    key = int(key)
    kind, rest = key // 10 ** 11, key % 10 ** 11
    return kind, rest // 10 ** 8, (rest // 10 ** 5) % 1000, rest % 10 ** 5


def kernel_name(key):
    """Decodes the library's profiling key (gemm.hip).  First digit = kernel family: 1 register-staged gemm_kernel
    (BM|BN|TA|TB|SPLIT|EPI), 2 register-staged dgrad+wgrad pair (BMdgrad|BMwgrad|EPI), 3 LDS-DMA forward kernel, 4 LDS-DMA kernel
    with transpose reads for k-major operands, 6 the dgrad+wgrad pair on that pipeline."""
    kind, b0, b1, tail = _key_fields(key)
    if kind == 7:
        return f'blk_attn_kernel<{tail}> (fused norm1 + qkv + attention, split3)'
    if kind == 8:
        return f'blk_mlp1_kernel<{tail}> (fused norm2 + fc1 + GELU, split3)'
    if kind == 9:
        return f'blk_attn_bwd_kernel<{tail}> (fused proj dgrad + attention backward)'
    if kind in R5_KERNELS:
        return R5_KERNELS[kind][0]
    epi = EPI_NAMES.get(tail % 100, tail % 100)
    ta, tb, sp = (tail // 10000) % 10, (tail // 1000) % 10, (tail // 100) % 10
    lay = f'{"T" if ta else "N"}{"T" if not tb else "N"}'
    if kind == 2:
        return f'gemm_pair_kernel<dgrad {b0}x64 NN {epi} || wgrad {b1}x64 TN atomic>'
    if kind == 6:
        return f'gemm_pair_dmat_kernel<dgrad {b0}x{b1} NN {epi} || wgrad {b0}x{b1} TN atomic; LDS-DMA + transpose reads>'
    if kind == 3 and b0 == 256 and b1 == 256:
        return f'gemm_nt_fat_kernel<256,256,NT,split3,{epi}> (eight waves)'
    if kind == 3:
        return f'gemm_nt_dma_kernel<{b0},{b1},NT,{"split3" if sp else "bf16"},{epi}>'
    if kind == 4 and b0 == 256 and b1 == 256:
        return f'gemm_nn_fat_kernel<256,256,{lay},bf16,{epi}> (eight waves)'
    if kind == 4:
        return f'gemm_dmat_kernel<{b0},{b1},{lay},bf16,{epi}>'
    return f'gemm_kernel<{b0},{b1},{lay},{"split3" if sp else "bf16"},{epi}>'


def rocprof_name(key):
    """The same kernel as rocprofv3 prints it (template arguments), for looking it up in profiles/*.json."""
    kind, b0, b1, tail = _key_fields(key)
    if kind in (7, 8):
        return f'{"blk_attn_kernel" if kind == 7 else "blk_mlp1_kernel"}<{tail}>'
    if kind == 9:
        return f'blk_attn_bwd_kernel<{tail}>'
    if kind in R5_KERNELS:
        return R5_KERNELS[kind][1]
    tf = lambda v: 'true' if v else 'false'
    ta, tb, sp, epi = (tail // 10000) % 10, (tail // 1000) % 10, (tail // 100) % 10, tail % 100
    if kind == 2:
        return f'gemm_pair_kernel<{b0}, {epi}, {b1}>'
    if kind == 6:
        return f'gemm_pair_dmat_kernel<{epi}, 3>'
    if kind == 3 and b0 == 256 and b1 == 256:
        return f'gemm_nt_fat_kernel<{epi}, 0>'
    if kind == 3:
        return f'gemm_nt_dma_kernel<{tf(sp)}, {epi}, {3 if b1 == 256 else 2}, {32 if (sp and b0 == 128) else 64}, {b0}, {b1}>'     # 128x256: three-stage ring
    if kind == 4 and b0 == 256 and b1 == 256:
        return f'gemm_nn_fat_kernel<{epi}>'
    if kind == 4:
        return f'gemm_dmat_kernel<{tf(ta)}, {tf(tb)}, {epi}, {2 if b0 == 128 else 3}, {b0}, {b1}>'
    return f'gemm_kernel<{b0}, {b1}, {tf(ta)}, {tf(tb)}, {tf(sp)}, {epi}>'


PMC_TRAFFIC_FILE = 'profiles/r05_pmc_step_traffic.json'           # cfg-2; main() switches to ..._pmc_<config>_traffic.json for the others


def pmc_traffic(key):
    """HBM-side bytes per launch of this kernel from the committed PMC passes (tools/pmc_step.sh -> profiles/), or None.
    bench.py cannot collect PMC counters itself (they need rocprofv3 around the process, one pass per counter group), so the figure
    is only as fresh as that file: it is matched by the EXACT rocprofv3 kernel name (a changed kernel has a changed template
    argument list or no entry -> traffic null, never a neighbour's figure), and the line carries the file's hash and the commit /
    date the passes were taken at."""
    import hashlib
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), PMC_TRAFFIC_FILE)
    try:
        raw = open(path, 'rb').read()
        doc = json.loads(raw)
        prov = {'file': PMC_TRAFFIC_FILE, 'sha16': hashlib.sha256(raw).hexdigest()[:16], 'collected_at_head': doc.get('head'),
                'collected_on': doc.get('date')}
        want = rocprof_name(key)
        stem = want[:-1]            # the leading template arguments identify the instantiation; later ones (ring depth, wave grid,
        hits = [n for n in doc['kernels']        # k-tail flag ...) were appended over the rounds -- exactly ONE entry may carry this stem
                if n == want or n.startswith(want + '(') or (n.startswith(stem) and n[len(stem):len(stem) + 1] == ',')]
        if len(hits) != 1:
            return None, None, prov
        k = doc['kernels'][hits[0]]
        return k['hbm_bytes_per_launch'], k, prov
    except (OSError, ValueError, KeyError):
        return None, None, None


def cpu_model():
    """Host CPU model string (lscpu's 'Model name'; /proc/cpuinfo as the fallback)."""
    try:
        import subprocess
        for line in subprocess.run(['lscpu'], capture_output=True, text=True, timeout=5).stdout.splitlines():
            if line.startswith('Model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(x, y, budget_s=24.0, pos_embedding='default', dropout=0.0, full_batch=None):
    """The oracle's training step (forward + CE + autograd backward + Adam on every used parameter) on host cores, at 8 threads, 32
    threads and every core torch sees; `value` is the BEST of the three, `cores` the thread count that gave it."""
    from oracle import voxel_oracle as vo
    kw = dict(backbone=CFG['backbone'], embed_layer=CFG['embed_layer'], cell=CFG['cell'], patch=CFG['patch'])
    if pos_embedding != 'default':
        kw.update(pos_embedding=pos_embedding)
        if dropout > 0:
            kw.update(training=True, dropout_p=dropout, hash_seed=1)
    all_threads = torch.get_num_threads()
    settings = sorted({t for t in (8, 32, all_threads) if t <= all_threads} | {all_threads})
    by_threads, best = {}, None
    try:
        for threads in settings:
            torch.set_num_threads(threads)
            sd = vo.init_state_dict(seed=9, voxel_size=CFG['voxel_size'], pos_embedding=pos_embedding,
                                    **{k: CFG[k] for k in ('backbone', 'embed_layer', 'cell', 'patch', 'n_classes')})
            names = vo.used_param_names(sd, pos_embedding)
            m = {k: torch.zeros_like(sd[k]) for k in names}
            v = {k: torch.zeros_like(sd[k]) for k in names}

            def one(step):
                _, loss, grads = vo.loss_and_grads(sd, x, y, **kw)
                for k, g in grads.items():
                    vo.adam_step(sd[k], g, m[k], v[k], step)
                return float(loss)

            share = budget_s / len(settings)
            t0 = time.perf_counter()
            one(1)                                             # warm-up (counted only if it already exhausts this setting's share)
            warm = time.perf_counter() - t0
            t0 = time.perf_counter()
            n, el = 0, 0.0
            while warm < share:
                one(n + 2)
                n += 1
                el = time.perf_counter() - t0
                if el > share - warm or n >= 50:
                    break
            if n == 0:
                n, el = 1, warm
            rate = n * x.shape[0] / el
            by_threads[str(threads)] = dict(value=round(rate, 3), steps=n, seconds=round(el, 2))
            if best is None or rate > best[0]:
                best = (rate, threads, n, el)
    finally:
        torch.set_num_threads(all_threads)
    rate, threads, n, el = best
    reduced = f' (REDUCED batch: the benchmark runs {full_batch} per GPU)' if full_batch and full_batch != x.shape[0] else ''
    return dict(value=round(rate, 3), unit='voxels/sec', cores=threads, kind='port', cpu_model=cpu_model(), host_threads=all_threads,
                by_threads=by_threads,
                sample=f'{n} full training steps (fwd+bwd+Adam) of the fp32 PyTorch-CPU oracle at batch {x.shape[0]}{reduced}, best of '
                       f'{"/".join(str(t) for t in settings)} threads = {threads} threads, {el:.1f} s')


def timed(fn, reps, dev_sync=True):
    """Average milliseconds of fn() over reps launches, HIP events on torch's current stream (= the library's launch stream)."""
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def rccl_identity(dev, world):
    """Proof that the collective spans `world` distinct devices: every rank contributes (device index, device uuid / bus id)."""
    import torch.distributed as dist
    props = torch.cuda.get_device_properties(dev)
    ident = f'{getattr(props, "uuid", "")}|{getattr(props, "pci_bus_id", "")}|{getattr(props, "pci_device_id", "")}|idx{dev.index}'
    if world == 1 or not dist.is_initialized():
        return dict(rccl_ranks=1, distinct_devices=1, backend=dist.get_backend() if dist.is_initialized() else None)
    got = [None] * world
    dist.all_gather_object(got, ident)
    return dict(rccl_ranks=dist.get_world_size(), distinct_devices=len(set(got)), backend=dist.get_backend())


def collective_diagnostics(trainer, step, step_ms, world, force, reps=10):
    """Per-bucket all-reduce wall time in isolation (events around the collective on the compute stream, ranks aligned by a barrier),
    the step time with the collectives suppressed, and from the two the fraction of the wire time that hid behind backward."""
    import torch.distributed as dist
    if not (world > 1 or force):
        return {}
    red = trainer.reducer
    ar_ms = []
    for (s0, e0) in red.slices:
        buf = red.flat[s0:e0]
        keep = buf.clone()

        def one():
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=red.group)
        if world > 1:
            dist.barrier()
        ar_ms.append(round(timed(one, reps), 4))
        buf.copy_(keep)
    graph_coll = bool(getattr(trainer, 'graph_collectives', False))      # (the point trainer launches its collectives from the host)
    if graph_coll:
        # the all-reduces are nodes of the step graph: the yardstick is the single-replica step (one graph, no collective at all)
        g1, sx1, sy1, _ = trainer.eng.capture_train_step(trainer._cap['B'])
        sx1.copy_(trainer._cap['x']); sy1.copy_(trainer._cap['y'])
        step_wo = g1.replay
    else:
        red.skip = True
        step_wo = step
    for _ in range(3):
        step_wo()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        step_wo()
    torch.cuda.synchronize()
    t_skip = (time.perf_counter() - t0) / n * 1e3
    red.skip = False
    t = torch.tensor([t_skip], dtype=torch.float64, device=red.flat.device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t_skip = float(t.item())
    total = sum(ar_ms)
    exposed = max(0.0, step_ms - t_skip)
    return dict(allreduce_ms=ar_ms, allreduce_bytes=[(e - s) * red.flat.element_size() for s, e in red.slices],
                allreduce_busbw_GBps=[round(2 * (world - 1) / max(world, 1) * (e - s) * red.flat.element_size() / (ms * 1e-3) / 1e9, 1)
                                      if ms > 0 else None for (s, e), ms in zip(red.slices, ar_ms)],
                ms_per_step_without_collectives=round(t_skip, 4), exposed_comm_ms=round(exposed, 4),
                overlap_frac=round(1.0 - min(1.0, exposed / total), 4) if total > 0 else None,
                note='allreduce_ms: each bucket alone on an otherwise idle GPU; overlap_frac = 1 - (step - step without collectives) / '
                     'sum(allreduce_ms); "without collectives" = ' + ('the single-replica one-graph step' if graph_coll else
                                                                      'the same segmented step with the collectives suppressed'))


def point_cpu_baseline(c, backbone, budget_s=18.0):
    """The point oracle's training step (forward incl. FPS / kNN, CE, autograd backward, SGD+momentum) on host cores, reduced batch, at 8
    threads, 32 threads and every core torch sees; `value` is the BEST of the three, `cores` the thread count that gave it (the voxel leg's
    protocol: an oversubscribed all-threads run is the worst of the three on the EPYC hosts of the pool)."""
    from oracle import point_oracle as po
    B = c['cpu_batch']
    x, y, starts = po.synthetic_points(B, c['n_points'], c['d_points'], c['n_classes'], c['task'], seed=9)
    all_threads = torch.get_num_threads()
    settings = sorted({t for t in (8, 32, all_threads) if t <= all_threads} | {all_threads})
    by_threads, best = {}, None
    try:
        for threads in settings:
            torch.set_num_threads(threads)
            sd = po.init_state_dict(backbone=backbone, n_classes=c['n_classes'], d_points=c['d_points'], seed=9)
            names = po.used_param_names(sd)
            buf = {k: torch.zeros_like(sd[k]) for k in names}

            def one(first):
                _, loss, grads, _ = po.loss_and_grads(sd, x, y, backbone=backbone, starts=starts, task=c['task'])
                for k, g in grads.items():
                    po.sgd_momentum_step(sd[k], g, buf[k], first=first)
                return float(loss)

            share = budget_s / len(settings)
            t0 = time.perf_counter()
            one(True)
            warm = time.perf_counter() - t0
            t0, n, el = time.perf_counter(), 0, warm
            while warm < share:
                one(False)
                n += 1
                el = time.perf_counter() - t0
                if el > share - warm or n >= 50:
                    break
            if n == 0:
                n, el = 1, warm
            rate = n * B / el
            by_threads[str(threads)] = dict(value=round(rate, 3), steps=n, seconds=round(el, 2))
            if best is None or rate > best[0]:
                best = (rate, threads, n, el)
    finally:
        torch.set_num_threads(all_threads)
    rate, threads, n, el = best
    return dict(value=round(rate, 3), unit='clouds/sec', cores=threads, kind='port', cpu_model=cpu_model(), host_threads=all_threads,
                by_threads=by_threads,
                sample=f'{n} full training steps (FPS/kNN + fwd + bwd + SGD) of the fp32 PyTorch-CPU oracle at batch {B} (REDUCED: the '
                       f'benchmark runs {c["batch"]} per GPU), best of {"/".join(str(t) for t in settings)} threads = {threads} threads, {el:.1f} s')


def main_points(args):
    """BASELINE.json configs[3] / configs[4]: one step = FPS/kNN geometry + forward + CE + backward + (all-reduce) + SGD(momentum)."""
    import ctypes
    import torch.distributed as dist
    import simple3d_former_amd as s3d                           # noqa: F401
    from simple3d_former_amd import _lib as L
    from simple3d_former_amd.point_engine import PointEngine, KNN
    from simple3d_former_amd.parallel import PointDataParallelTrainer
    from oracle import point_oracle as po                      # synthetic-input recipe + cpu_baseline leg only

    c = POINT_CONFIGS[args.config]
    backbone = 'deit_tiny_patch16_224'
    B = args.batch or c['batch']
    world, rank, local_rank = (int(os.environ.get(k, d)) for k, d in (('WORLD_SIZE', '1'), ('RANK', '0'), ('LOCAL_RANK', '0')))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'
    dev_index = int(os.environ.get('S3D_BENCH_DEVICE', local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1 or args.force_collectives:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29541')
        dist.init_process_group(backend=os.environ.get('S3D_BENCH_BACKEND', 'nccl'), init_method='env://', world_size=world, rank=rank)
    eng = PointEngine(backbone=backbone, n_points=c['n_points'], d_points=c['d_points'], n_classes=c['n_classes'], task=c['task'],
                      device=dev, lr=0.01, momentum=0.9)       # config/cls.yaml:3,6; train_cls.py:91
    eng.load_state_dict(po.init_state_dict(backbone=backbone, n_classes=c['n_classes'], d_points=c['d_points'], seed=9))
    x, y, starts = po.synthetic_points(B, c['n_points'], c['d_points'], c['n_classes'], c['task'], seed=9 + rank)
    x, y, starts = x.to(dev), y.to(dev), tuple(t.to(dev) for t in starts)
    ident = rccl_identity(dev, world)
    dp = world > 1 or args.force_collectives
    pipelined = not args.no_pipeline and not args.no_graphs

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(); torch.cuda.synchronize()

    if dp:
        tr = PointDataParallelTrainer(eng, use_graphs=not args.no_graphs, force_collectives=args.force_collectives)
        if pipelined:
            tr.prime(x, y, starts)
            step = lambda: tr.step_pipelined(x, y, starts)
            launch = 'hipGraph replay (3 graphs + 2 host-launched all-reduces), geometry (FPS/kNN) one step ahead on a side stream'
        else:
            loss_t = tr.step(x, y, starts)
            step = tr.step_graph if not args.no_graphs else (lambda: tr.step_eager(x, y, starts))
            launch = ('hipGraph replay (3 graphs + 2 host-launched all-reduces)' if not args.no_graphs else 'eager')
    elif pipelined:
        xs, ys, sts = [x, x.clone()], [y, y.clone()], [starts, tuple(t.clone() for t in starts)]
        graphs, loss_t = eng.capture_train_step_pipelined(xs, ys, sts)
        state = {'p': 0}

        def step():
            graphs[state['p']].replay(); state['p'] ^= 1
        launch = 'hipGraph replay: two chain graphs per step on two streams (geometry FPS/kNN of the next batch | training step)'
    elif not args.no_graphs:
        graph, loss_t = eng.capture_train_step(x, y, starts)
        step = graph.replay
        launch = 'hipGraph replay'
    else:
        step = lambda: eng.train_step(x, y, starts)
        loss_t = eng.workspace(B).loss[0:1]
        launch = 'eager'
    first_loss = None
    for i in range(max(args.warmup, 1)):
        step()
        if i == 0:
            first_loss = float(eng.workspace(B).loss[0])
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    last_loss = float(eng.workspace(B).loss[0])
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed
    out = {'metric': c['metric'], 'value': round(value, 1), 'unit': 'clouds/sec', 'n_gpus': world, 'steps': args.steps,
           'warmup': args.warmup, 'ms_per_step': round(ms, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
           'dtype': 'bf16',
           'precision_note': 'bf16 MFMA operands (split-bf16 forward, plain bf16 backward), fp32 accumulation / BatchNorm (fp64 column sums) / '
                             'LayerNorm / softmax / loss / SGD; FPS, kNN and 3-NN indices bit-exact vs the fp32 reference',
           'data': f'synthetic (seeded unit-ball clouds with unit normals{" + one-hot(16) object label" if c["d_points"] > 6 else ""}, '
                   'random-init weights of the reference architecture, FPS start draws as explicit inputs)',
           'config': {'workload': c['workload'], 'batch_per_gpu': B, 'global_batch': world * B, 'points_per_cloud': c['n_points'],
                      'parallelism': f'dp{world}', 'launch': launch,
                      'collectives': 'none' if not dp else 'host-launched between graph segments (fp32, 2 buckets)'},
           'points_per_sec': round(value * c['n_points'], 0),
           'rccl_ranks': ident['rccl_ranks'], 'distinct_devices': ident['distinct_devices'], 'dist_backend': ident['backend'],
           'loss_first_step': round(first_loss, 5), 'loss_last_step': round(last_loss, 5)}
    if c['task'] == 'seg':
        out['seg_points_per_sec'] = out['points_per_sec']
    if dp and not args.no_diagnostics:
        out.update(collective_diagnostics(tr, step, ms, world, args.force_collectives))
    if rank == 0 and not args.no_roofline:
        # ---- HBM-bound side: the non-GEMM operators of TransitionDown 0, timed live on this step's own tensors (HIP events on the
        # launch stream, 20 launches each).  Algorithmic bytes = what the operator must move once: inputs read once,
        # outputs written once -- re-reads (e.g. the statistics pass of BatchNorm) count against the achieved rate.
        ws = eng.workspace(B)
        t0_, lay = ws.td[0], eng.td[0]
        R, ch, G = t0_.R, eng.ch[0], B * t0_.S
        hbm = []

        def add(name, fn, nbytes, reps=20, note=None):
            ms_ = timed(fn, reps)
            hbm.append(dict(kernel=name, avg_us=round(ms_ * 1e3, 2), bytes_per_launch=int(nbytes),
                            achieved=round(nbytes / (ms_ * 1e-3) / 1e9, 1), frac=round(nbytes / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                            **({'note': note} if note else {})))
        eng.training = True
        keep = [b.clone() for b in eng.bn_buffers()]
        add('batchnorm_bwd (td0.mlp_bns.1: relu/max backward + stats + apply; s3d_batchnorm_bwd)',
            lambda: lay['b1'].bwd(t0_.x2, R, t0_.dout, t0_.dx, K=KNN, arg=t0_.arg), R * ch * (4 + 2) + G * ch * (4 + 1))
        add('batchnorm_bwd (td0.mlp_bns.0: relu backward + stats + apply)',
            lambda: lay["b0"].bwd(t0_.x1, R, t0_.dy1, t0_.dx), R * ch * (4 + 2 + 2))        # x fp32, dy bf16 (from the dgrad epilogue), dx bf16
        add('batchnorm_fwd + relu + max over 16 neighbours (td0.mlp_bns.1; s3d_batchnorm_fwd)',
            lambda: lay['b1'].fwd(t0_.x2, R, K=KNN, y=t0_.out, arg=t0_.arg), R * ch * 4 + G * ch * (4 + 1))
        add('group_project_fwd (td0: Pf[idx] + xyz_rel.Wx^T + b, BatchNorm column sums fused)',
            lambda: L.check(eng.lib.s3d_group_project_fwd(ctypes.byref(lay['gp'].args(t0_, t0_.xyz_in, B, Pf=t0_.Pf, x=t0_.x1, ldx=ch,
                                                                                        sums=lay['b0'].sums)), L.current_stream()), 'gp'),
            R * ch * 4 + B * t0_.Nin * ch * 4 + R * 4)
        g0 = ws.geo[0].td[0]
        add('fps (level 0: one workgroup per cloud, npoint sequential rounds; latency-bound by construction)',
            lambda: L.check(eng.lib.s3d_fps(L.ptr(ws.geo[0].xyz), ctypes.c_long(3), L.ptr(starts[0]), B, t0_.Nin, t0_.S, L.ptr(g0['fps_idx']),
                                            L.ptr(g0['new_xyz']), L.current_stream()), 'fps'),
            B * (t0_.Nin * 12 + t0_.S * 16), reps=5, note='HBM-minimal bytes N*12 in + npoint*16 out per cloud (SURVEY 8(d)); the kernel is a chain of npoint dependent arg-max rounds')
        add('knn16 (level 0)',
            lambda: L.check(eng.lib.s3d_knn(L.ptr(g0['new_xyz']), L.ptr(ws.geo[0].xyz), B, t0_.S, t0_.Nin, KNN, L.ptr(g0['idx']), None,
                                            L.current_stream()), 'knn'),
            B * (t0_.Nin * 12 + t0_.S * 12 + t0_.S * KNN * 4), reps=5)
        for b_, k_ in zip(eng.bn_buffers(), keep):
            b_.copy_(k_)
        dom = max(hbm[:4], key=lambda k: k['avg_us'])          # dominant bandwidth-bound operator (FPS / kNN are latency chains)
        # ---- MFMA side: GEMM launches of an instrumented eager step
        lib = L.lib()
        lib.s3d_prof_enable(1)
        n_inst = 3
        for _ in range(n_inst):
            eng.train_step(x, y, starts)
        torch.cuda.synchronize()
        rows = (ctypes.c_double * (4 * 64))()
        n = lib.s3d_prof_collect(rows, 64)
        lib.s3d_prof_enable(0)
        ov = ctypes.c_double(0.0)
        lib.s3d_prof_event_overhead(L.current_stream(), ctypes.byref(ov))
        ks = [(rows[4 * i], rows[4 * i + 1], max(rows[4 * i + 2] - rows[4 * i + 1] * ov.value * 1e-3, 1e-9), rows[4 * i + 3]) for i in range(min(n, 64))]
        mf = None
        if ks:
            d = max(ks, key=lambda k: k[2])
            ach = d[3] / (d[2] * 1e-3) / 1e12
            mf = dict(bound='mfma', kernel=kernel_name(d[0]), achieved=round(ach, 2), peak=MFMA_BF16_PEAK_TFLOPS, unit='TFLOP/s',
                      frac=round(ach / MFMA_BF16_PEAK_TFLOPS, 5), avg_launch_us=round(d[2] / d[1] * 1e3, 3),
                      launches_per_step=round(d[1] / n_inst, 1), flops_per_launch=round(d[3] / d[1], 0),
                      mfma_issue_factor=3 if 'split3' in kernel_name(d[0]) else 1,
                      timing='HIP events on the launch stream around every GEMM launch of an instrumented eager step, minus the empty-bracket time',
                      all_gemm_kernels=dict(achieved=round(sum(k[3] for k in ks) / (sum(k[2] for k in ks) * 1e-3) / 1e12, 2),
                                            ms_per_step=round(sum(k[2] for k in ks) / n_inst, 4)))
        out['roofline'] = dict(bound='hbm', kernel=dom['kernel'], achieved=dom['achieved'], peak=HBM_PEAK_GBPS, unit='GB/s', frac=dom['frac'],
                               traffic=None, avg_launch_us=dom['avg_us'], bytes_per_launch=dom['bytes_per_launch'],
                               timing='HIP events on the launch stream around 20 back-to-back launches of the operator on the step\'s own tensors',
                               per_kernel=hbm, mfma=mf)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = point_cpu_baseline(c, backbone)
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    if world > 1:
        barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or args.force_collectives:
        if world > 1:
            barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-graphs', action='store_true', help='eager launches instead of HIP-graph replay')
    ap.add_argument('--plain-bf16', action='store_true', help='one-MFMA forward (fails the 1e-3 logit bar; for comparison only)')
    ap.add_argument('--backward', choices=['bf16', 'split'], default='bf16',
                    help="backward precision of the voxel configs: bf16 (default: plain bf16 MFMA operands -- asserted against the reference's own "
                         "seed-to-seed spread of trained accuracy / final loss, tests/test_gpu_trajectory.py) or split (every gradient product on "
                         "hi + lo operands, three MFMAs: VoxelEngine(precise_backward=True), the gradient-parity mode)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--force-collectives', action='store_true',
                    help='diagnostic: run the segmented multi-GPU step (RCCL calls between graph segments) even at N=1')
    ap.add_argument('--graph-collectives', choices=['auto', 'on', 'off'], default='auto', nargs='?', const='on',
                    help='one HIP graph per step with the RCCL all-reduces captured inside.  auto (default): after a child-process preflight '
                         'on the same ranks / devices has shown that captured all-reduces replay correctly, else one graph per backward '
                         'segment with the collectives launched from the host in between; on: no preflight; off: never')
    ap.add_argument('--buckets', type=int, default=None, help='gradient buckets (default: 4 with captured collectives, 2 with host-launched ones)')
    ap.add_argument('--config', choices=sorted(CONFIGS) + sorted(POINT_CONFIGS), default='cfg2')
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch override (non-headline experiments)')
    ap.add_argument('--wire', choices=['auto', 'fp32', 'bf16'], default='auto',
                    help="gradient all-reduce format: fp32 (DDP's arithmetic; auto = fp32) or bf16 (half the xGMI bytes, narrower than the reference)")
    ap.add_argument('--single-update', action='store_true',
                    help='N > 1 path: wait for every bucket, then ONE Adam launch (rounds 1 - 4) instead of Adam bucket by bucket as the all-reduces finish')
    ap.add_argument('--bucket-blocks', type=int, default=None, help='uniform gradient buckets of this many blocks (default: geometric, --buckets)')
    ap.add_argument('--standin-latency-us', type=float, default=0.0, help='fixed start-up cost added to every stand-in collective')
    ap.add_argument('--standin-gbps', type=float, default=0.0,
                    help='one-rank diagnostic (with --force-collectives): every bucket all-reduce is replaced by a copy kernel of that bus bandwidth on a '
                         'side stream / graph branch -- what a LIVE collective branch costs the captured step (profiles/r05_dp_branch_tax.txt)')
    ap.add_argument('--dp', choices=['auto', 'sharded', 'replicated'], default='auto',
                    help='N > 1 design of the voxel configs.  sharded (auto at N > 1): per bucket reduce-scatter -> Adam on the local 1/N shard -> '
                         'all-gather overlapped with the NEXT forward (parallel.ShardedDataParallelTrainer); replicated: bucketed all-reduce + the '
                         'full Adam on every rank (rounds 1 - 5, parallel.DataParallelTrainer)')
    ap.add_argument('--bucket-list', type=str, default=None, help='sharded design: blocks per bucket in backward order, e.g. 3,3,3,2,1 (the default at depth 12)')
    ap.add_argument('--emulate-world', type=int, default=0,
                    help='one-rank diagnostic of the sharded design (with --standin-gbps): collectives replaced by copy kernels of an E-rank '
                         "ring's reduce-scatter / all-gather duration, Adam on 1/E of every bucket; timing only (profiles/r06_dp_sharded.txt)")
    ap.add_argument('--event-graph', action='store_true',
                    help='ONE graph with an event-record node behind every backward segment, collectives launched from a side stream on '
                         'those events (measured slower than the default on this runtime: one graph per segment, collectives in between)')
    ap.add_argument('--no-diagnostics', action='store_true', help='skip the per-bucket all-reduce / overlap measurement at N > 1')
    ap.add_argument('--no-pipeline', action='store_true',
                    help='point configs: FPS / kNN inside the step instead of one step ahead on the side stream (the default)')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` the way `--gpus 1` is run: re-launch this very command line as N ranks on this node (one process per
        # GPU over RCCL), rendezvous on 127.0.0.1 at a free port; the ranks' stdout is ours, so rank 0's JSON line is the last line printed
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        return sys.exit(subprocess.call(cmd))
    if args.config in POINT_CONFIGS:
        return main_points(args)
    global CFG, BATCH_PER_GPU, TRAIN_FLOPS_PER_SAMPLE, PMC_TRAFFIC_FILE
    conf = CONFIGS[args.config]
    for rnd in ('r06', 'r05', 'r04', 'r03'):                  # PMC_BENCH_ARGS="--config cfg3 .." tools/pmc_step.sh; the newest passes that exist
        PMC_TRAFFIC_FILE = f'profiles/{rnd}_pmc_step_traffic.json' if args.config == 'cfg2' else f'profiles/{rnd}_pmc_{args.config}_traffic.json'
        if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), PMC_TRAFFIC_FILE)):
            break
    CFG, BATCH_PER_GPU, TRAIN_FLOPS_PER_SAMPLE = conf['cfg'], args.batch or conf['batch'], conf['train_flops']

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
    eng = s3d.VoxelEngine(device=dev, split=not args.plain_bf16, pos_embedding=conf['pos_embedding'], backward=args.backward, **CFG)
    sd = vo.init_state_dict(seed=9, pos_embedding=conf['pos_embedding'], **CFG)
    eng.load_state_dict(sd)
    x_cpu, y_cpu = vo.synthetic_batch(BATCH_PER_GPU, CFG['voxel_size'], CFG['n_classes'], seed=9 + rank)
    x, y = x_cpu.to(dev), y_cpu.to(dev)
    if conf.get('dropout') and os.environ.get('S3D_BENCH_NO_DROPOUT') != '1':          # (tuning aid: the cost of the dropout masks)
        eng.set_dropout(conf['dropout'], seed=9)                # model.train(): nn.TransformerEncoderLayer(dropout=0.1)
    # fp32 on the wire = what the reference's DDP all-reduces (train_cls_voxel.py:155-159); bf16 (half the xGMI bytes) is opt-in
    wire = 'fp32' if args.wire == 'auto' else args.wire
    sharded = args.dp == 'sharded' or (args.dp == 'auto' and (world > 1 or args.emulate_world > 0))
    if sharded:
        from simple3d_former_amd.parallel import ShardedDataParallelTrainer
        eng.set_optimizer(lr=1e-3)                              # README recipe (README.md:60)
        trainer = ShardedDataParallelTrainer(eng, bucket_blocks=[int(v) for v in args.bucket_list.split(',')] if args.bucket_list else None,
                                             use_graphs=not args.no_graphs, force_collectives=args.force_collectives,
                                             graph_collectives={'auto': 'auto', 'on': True, 'off': False}[args.graph_collectives],
                                             standin_gbps=args.standin_gbps, standin_latency_us=args.standin_latency_us,
                                             emulate_world=args.emulate_world)
        args.no_diagnostics = True                              # (the per-bucket all-reduce diagnostics belong to the replicated design)
    else:
        trainer = DataParallelTrainer(eng, n_buckets=args.buckets, use_graphs=not args.no_graphs,
                                      force_collectives=args.force_collectives,
                                      graph_collectives={'auto': 'auto', 'on': True, 'off': False}[args.graph_collectives], wire=wire,
                                      event_graph=args.event_graph, sliced_adam=not args.single_update, standin_gbps=args.standin_gbps,
                                      blocks_per_bucket=args.bucket_blocks, standin_latency_us=args.standin_latency_us)
        trainer.set_optimizer(lr=1e-3)                          # README recipe (README.md:60)
    ident = rccl_identity(dev, world)
    if world > 1:
        assert ident['rccl_ranks'] == world and ident['distinct_devices'] == world or os.environ.get('S3D_BENCH_DEVICE') is not None, \
            f'the process group does not span {world} distinct GPUs: {ident}'

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
    if sharded:
        trainer.sync_parameters()       # the last step's all-gather phase (otherwise the next step's first phase): inside the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    L.lib().s3d_prof_skip_get.restype = ctypes.c_double
    suppressed = L.lib().s3d_prof_skip_get() != 0.0           # difference timing (roofline leg, below) must not be active in the timed loop
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
                           'bf16 MFMA operands: split-bf16 (hi+lo, 3 MFMAs per product) forward, '
                           + ('split-bf16 backward too (gradient-parity mode)' if args.backward == 'split' else 'plain bf16 backward')
                           + '; fp32 accumulation, residual stream, LayerNorm/softmax/GELU, loss and Adam'),
        'backward_precision': eng.backward_precision,
        'data': f'synthetic (seeded 10%-occupancy {CFG["voxel_size"]}^3 grids, random-init weights of the reference architecture)',
        'config': {'workload': conf['workload'], 'batch_per_gpu': BATCH_PER_GPU,
                   'global_batch': world * BATCH_PER_GPU, 'tokens_per_sample': eng.ntok, 'parallelism': f'dp{world}',
                   'launch': 'eager' if args.no_graphs else 'hipGraph replay',
                   'grad_buckets': len(trainer.slices), 'grad_wire': wire,
                   'dp_design': ('sharded optimizer: reduce-scatter -> Adam on the 1/N shard -> all-gather overlapped with the next forward'
                                 if sharded else 'replicated optimizer: bucketed all-reduce overlapped with backward'),
                   'bucket_blocks': [f - l + 1 for f, l in trainer.segments] if sharded else None,
                   'emulate_world': getattr(trainer, 'emulate_world', 0) or None,
                   'collectives': trainer.collectives_mode()},
        'rccl_ranks': ident['rccl_ranks'], 'distinct_devices': ident['distinct_devices'], 'dist_backend': ident['backend'],
        'voxel_cells_per_sec': round(value * CFG['voxel_size'] ** 3, 0),
        'algorithmic_tflops': round(value * TRAIN_FLOPS_PER_SAMPLE / 1e12, 2),
        'loss_first_step': round(first_loss, 5) if first_loss is not None else None, 'loss_last_step': round(final_loss, 5),
        'loss_note': 'one repeated batch at lr 1e-3: plateau at 3.385 (label-histogram entropy) from step ~20, chaotic escape at step 160 - 260+ '
                     '(CPU oracle: ~168), then ~1e-3 -- profiles/r06_single_batch_dynamics.txt' if args.config == 'cfg2' else None,
        'kernels_suppressed': bool(suppressed),
    }

    if trainer.preflight is not None:
        out['graph_collectives_preflight'] = {'ok': bool(trainer.preflight[0]), 'detail': trainer.preflight[1]}
    if (world > 1 or args.force_collectives) and not args.no_diagnostics:
        out.update(collective_diagnostics(trainer, step, ms, world, args.force_collectives))   # every rank takes part
    if rank == 0 and not args.no_roofline:
        # instrumented eager pass: HIP events around every GEMM launch, on the launch stream
        lib = L.lib()
        n_inst = max(3, min(10, args.steps))
        lib.s3d_prof_enable(1)
        for _ in range(n_inst):
            eng.train_step(x, y)          # engine-only step (no collective): rank-0 kernel timing
        torch.cuda.synchronize()
        pass  # (ctypes: module-level import)
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
            traffic, tdetail, tprov = pmc_traffic(dom[0])
            ev_by_key = {k[0]: k for k in ev}
            out['roofline'] = {
                'bound': 'mfma', 'kernel': kernel_name(dom[0]), 'achieved': round(achieved, 2), 'peak': MFMA_BF16_PEAK_TFLOPS,
                'unit': 'TFLOP/s', 'frac': round(achieved / MFMA_BF16_PEAK_TFLOPS, 5), 'traffic': traffic,
                'traffic_unit': 'bytes per launch (memory side of L2: HBM + Infinity Cache)',
                'traffic_source': ('rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE, separate passes; kernel '
                                   + rocprof_name(dom[0])) if traffic else None,
                'traffic_provenance': tprov,
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
        cb = conf.get('cpu_batch') or BATCH_PER_GPU             # cfg-3: one 128^3 sample is ~2 TFLOP of fp32 CPU work per step
        out['cpu_baseline'] = cpu_baseline(x_cpu[:cb], y_cpu[:cb], pos_embedding=conf['pos_embedding'], dropout=conf.get('dropout', 0.0),
                                           full_batch=BATCH_PER_GPU)

    # RCCL writes a version banner to the C stdout of every rank, block-buffered until the process exits -- i.e. AFTER a JSON line
    # printed from Python.  Push it out on every rank first, so that the JSON line is the last thing on stdout.

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
