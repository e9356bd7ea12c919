"""CPU, world_size = 2 over gloo: the data-parallel layer (parallel.py) -- parameter broadcast, bucketed gradient
all-reduce launched segment by segment during backward, 1/world averaging folded into the optimizer, and the
DistributedSampler index rule.  The HIP engine is replaced by a tiny CPU stand-in that implements the same interface,
so the test checks the DP protocol itself: two ranks on half batches == one process on the full batch."""
import contextlib
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from simple3d_former_amd.parallel import (BucketedGradReducer, DataParallelTrainer, PointDataParallelTrainer, ShardedDataParallelTrainer,
                                          broadcast_parameters, shard_indices)


class _Arena:
    def __init__(self, depth, width, seed):
        g = torch.Generator().manual_seed(seed)
        self.offsets, off = {}, 0
        for i in range(depth):
            self.offsets[f'blocks.{i}.norm1.weight'] = off
            off += width * width
        self.numel = off
        self.p = torch.randn(off, generator=g) * 0.3
        self.g = torch.zeros(off)
        self.m = torch.zeros(off)


class FakeEngine:
    """depth x (width x width) tanh layers, MSE-style loss; same interface as VoxelEngine for the DP trainer."""

    def __init__(self, depth=6, width=8, seed=0):
        self.depth, self.width = depth, width
        self.arena = _Arena(depth, width, seed)
        self.grad_scale, self.lr = 1.0, 0.1
        self.launch_log = []
        self.dropout_seed = torch.zeros(1, dtype=torch.int64)

    def W(self, i, flat=None):
        flat = self.arena.p if flat is None else flat
        o = self.arena.offsets[f'blocks.{i}.norm1.weight']
        return flat[o:o + self.width ** 2].view(self.width, self.width)

    capture_epoch = 0

    @contextlib.contextmanager
    def owning_grads(self):                                     # VoxelEngine.owning_grads: the trainer's backward owns the gradient arena
        self.owned_scopes = getattr(self, 'owned_scopes', 0) + 1
        yield

    def refresh_weight_planes(self): pass
    def advance_dropout_seed(self): self.dropout_seed += 1

    def set_optimizer(self, lr=None, betas=None, eps=None, grad_scale=None):
        if lr is not None: self.lr = lr
        if grad_scale is not None: self.grad_scale = grad_scale

    def pack_grads(self, start, end, wire):                     # s3d_pack_bf16: round-to-nearest-even, as torch's cast
        wire[start:end] = self.arena.g[start:end].to(torch.bfloat16)

    def grad_buckets(self, n):
        from simple3d_former_amd.engine import VoxelEngine
        return VoxelEngine.grad_buckets(self, n)

    def forward(self, x):
        self.acts = [x]
        for i in range(self.depth):
            self.acts.append(torch.tanh(self.acts[-1] @ self.W(i).t()))
        return self.acts[-1]

    def forward_loss(self, x, y, weight=None):
        self.forward(x)
        return self.cross_entropy(x.shape[0], y, weight)

    def cross_entropy(self, B, y, weight=None):
        self.dout = (self.acts[-1] - y) / B                     # d/dx of 0.5*mean_b |x-y|^2
        return 0.5 * ((self.acts[-1] - y) ** 2).sum() / B

    def backward(self, B, segments=None, on_segment=None):
        d = self.dout
        for si, (first, last) in enumerate(segments):
            for i in range(first, last - 1, -1):
                d = d * (1 - self.acts[i + 1] ** 2)
                self.W(i, self.arena.g).add_(d.t() @ self.acts[i])
                d = d @ self.W(i)
            if on_segment:
                on_segment(si)

    def adam_step(self, zero_grad=True, wire=None):             # plain SGD is enough to test the DP protocol
        g = self.arena.g if wire is None else wire.float()      # s3d_adam_step_wire: the update reads the bf16 sum
        self.arena.p.add_(g, alpha=-self.lr * self.grad_scale)
        if zero_grad:
            self.arena.g.zero_()

    # the sliced update (s3d_adam_begin / s3d_adam_apply): per-step bookkeeping once, then one arena slice at a time
    def adam_begin(self):
        self.update_log = getattr(self, 'update_log', []) + ['begin']

    def adam_apply(self, start, end, zero_grad=True, max_workgroups=0, wire=None):
        g = self.arena.g[start:end] if wire is None else wire[start:end].float()
        self.arena.p[start:end].add_(g, alpha=-self.lr * self.grad_scale)
        if zero_grad:
            self.arena.g[start:end].zero_()
        self.update_log.append((start, end))

    def adam_end(self):
        self.update_log.append('end')


def _init_group(rank, world, port):
    """gloo rendezvous through a file store (no port chosen ahead of time: no EADDRINUSE race between _free_port() and the bind)."""
    import datetime
    import tempfile
    path = os.path.join(tempfile.gettempdir(), f's3d_rdzv_{os.getppid()}_{port}')
    dist.init_process_group('gloo', init_method=f'file://{path}', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))



def _to_wire(obj):
    """Tensors cross the multiprocessing queue as numpy arrays (pickled by VALUE).  A torch tensor travels as a shared-memory handle that the parent
    has to fetch from the worker: when the worker has already exited, q.get() dies with EOFError -- a once-in-twenty-runs flake of this file."""
    if isinstance(obj, torch.Tensor):
        return ('__tensor__', obj.detach().cpu().numpy())
    if isinstance(obj, (tuple, list)):
        return type(obj)(_to_wire(o) for o in obj)
    return obj


def _from_wire(obj):
    if isinstance(obj, tuple) and len(obj) == 2 and isinstance(obj[0], str) and obj[0] == '__tensor__':
        return torch.from_numpy(obj[1].copy())
    if isinstance(obj, (tuple, list)):
        return type(obj)(_from_wire(o) for o in obj)
    return obj


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q, wire='fp32'):
    _init_group(rank, world, port)
    try:
        torch.manual_seed(100)
        X = torch.randn(8, 8); Y = torch.randn(8, 8)            # global batch, identical on every rank
        eng = FakeEngine(seed=rank)                              # DIFFERENT initial params per rank -> broadcast must fix it
        eng.lr = 0.05                                            # the trainer must keep the engine's own hyper-parameters
        tr = DataParallelTrainer(eng, n_buckets=3, use_graphs=False, wire=wire)
        assert tr.world == world and len(tr.slices) == 3 and eng.lr == 0.05 and eng.grad_scale == 0.5
        tr.set_optimizer(lr=0.1)                                 # ... and a later lr change must not reset 1/world
        assert eng.lr == 0.1 and eng.grad_scale == 0.5
        assert int(eng.dropout_seed) == 1000003 * rank           # per-rank dropout streams
        sl = slice(rank * 4, rank * 4 + 4)                       # contiguous half of the global batch
        for _ in range(3):
            tr.step_eager(X[sl], Y[sl])
        # raw reducer on its own flat tensor
        flat = torch.full((10,), float(rank + 1))
        red = BucketedGradReducer(flat, [(6, 10), (0, 6)])
        red.launch(0); red.launch(1); red.wait()
        assert int(eng.dropout_seed) == 1000003 * rank + 3       # advanced once per step
        # optimizer.step() ran bucket by bucket, in the order the buckets' collectives were launched, once per step
        assert tr.sliced_adam and eng.update_log == (['begin'] + [tuple(sl_) for sl_ in tr.slices] + ['end']) * 3, eng.update_log
        q.put(_to_wire((rank, eng.arena.p.clone(), flat.clone(), tr.segments, tr.slices)))
    finally:
        dist.destroy_process_group()


def test_two_rank_data_parallel_equals_single_process_full_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_from_wire(q.get(timeout=120)) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference: rank 0's initial params, full batch
    torch.manual_seed(100)
    X = torch.randn(8, 8); Y = torch.randn(8, 8)
    ref = FakeEngine(seed=0)
    segs, _ = ref.grad_buckets(1)
    for _ in range(3):
        ref.forward(X); ref.cross_entropy(8, Y); ref.backward(8, segments=segs); ref.adam_step()
    p0, p1 = res[0][1], res[1][1]
    assert torch.equal(p0, p1), 'replicas diverged'
    assert float((p0 - ref.arena.p).abs().max()) < 1e-6, 'DP on two half batches != full batch'
    assert torch.equal(res[0][2], torch.full((10,), 3.0))       # 1 + 2 summed over both buckets
    assert res[0][3] == [(5, 3), (2, 1), (0, 0)]                 # backward order, 3 buckets shrinking towards the input
    assert res[0][4] == [(192, 384), (64, 192), (0, 64)]         # contiguous arena slices tiling [0, numel)


def test_two_rank_bf16_wire_format_stays_within_the_rounding_bound():
    """wire='bf16': every bucket is rounded to bf16, summed over the ranks in bf16, and the update reads that sum.  Replicas must
    stay bitwise equal, and the parameters must track the fp32 single-process run within the bf16 rounding of the gradients
    (|delta p| <= lr * steps * 2^-8 * max|g| with two roundings per element: the pack and the bf16 sum)."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, 'bf16')) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_from_wire(q.get(timeout=120)) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(100)
    X = torch.randn(8, 8); Y = torch.randn(8, 8)
    ref = FakeEngine(seed=0)
    segs, _ = ref.grad_buckets(1)
    gmax = 0.0
    for _ in range(3):
        ref.forward(X); ref.cross_entropy(8, Y); ref.backward(8, segments=segs)
        gmax = max(gmax, float(ref.arena.g.abs().max()))
        ref.adam_step()
    p0, p1 = res[0][1], res[1][1]
    assert torch.equal(p0, p1), 'replicas diverged'
    err = float((p0 - ref.arena.p).abs().max())
    assert 0 < err <= 0.1 * 3 * 2 ** -8 * gmax * 1.5, (err, gmax)      # differs (it IS rounded), but only by the rounding


class FakePointEngine(FakeEngine):
    """Same stand-in with PointEngine's DP interface: backward in two halves around grad_split, running-stat buffers, SGD."""

    def __init__(self, seed=0):
        super().__init__(depth=6, width=8, seed=seed)
        self.stats = torch.full((4,), float(seed))               # "BatchNorm running statistics": replica-local after the broadcast
        self.buf = torch.zeros_like(self.arena.p)
        self.sgd_steps = torch.zeros(1, dtype=torch.int32)
        self.ws = type('WS', (), {})()
        self.ws.loss = torch.zeros(2)

    def bn_buffers(self): return [self.stats]
    def grad_split(self): return self.arena.offsets['blocks.2.norm1.weight']     # layers 2..5 = "top", 0..1 = "bottom"
    def workspace(self, B): return self.ws

    def forward(self, x, starts):
        self.stats.add_(float(x.sum()))                          # batch-dependent, never all-reduced
        return FakeEngine.forward(self, x)

    def cross_entropy(self, B, y):
        self.ws.loss[0] = FakeEngine.cross_entropy(self, B, y)
        return self.ws.loss[0]

    def backward_top(self, B):
        self._d = None
        FakePointEngine.backward(self, B, segments=[(5, 2)])

    def backward(self, B, segments=None, on_segment=None):       # keeps the running upstream gradient between the halves
        d = self.dout if getattr(self, '_d', None) is None else self._d
        for first, last in segments:
            for i in range(first, last - 1, -1):
                d = d * (1 - self.acts[i + 1] ** 2)
                self.W(i, self.arena.g).add_(d.t() @ self.acts[i])
                d = d @ self.W(i)
        self._d = d

    def backward_bottom(self, B):
        FakePointEngine.backward(self, B, segments=[(1, 0)])

    def sgd_step(self):
        self.arena.p.add_(self.arena.g, alpha=-self.lr * self.grad_scale)
        self.arena.g.zero_()


def _point_worker(rank, world, port, q):
    _init_group(rank, world, port)
    try:
        torch.manual_seed(100)
        X = torch.randn(8, 8); Y = torch.randn(8, 8)
        eng = FakePointEngine(seed=rank + 1)
        tr = PointDataParallelTrainer(eng, use_graphs=False)
        stats0 = eng.stats.clone()                               # after the constructor broadcast
        sl = slice(rank * 4, rank * 4 + 4)
        losses = [float(tr.step(X[sl], Y[sl], ())) for _ in range(3)]
        q.put(_to_wire((rank, eng.arena.p.clone(), stats0, eng.stats.clone(), tr.slices, eng.grad_scale, losses)))
    finally:
        dist.destroy_process_group()


def test_point_trainer_two_ranks_equal_single_process_full_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_point_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_from_wire(q.get(timeout=120)) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(100)
    X = torch.randn(8, 8); Y = torch.randn(8, 8)
    ref = FakePointEngine(seed=1)                                # rank 0's initial parameters, full batch, no process group
    tr = PointDataParallelTrainer(ref, use_graphs=False)
    assert tr.world == 1 and ref.grad_scale == 1.0
    for _ in range(3):
        tr.step(X, Y, ())
    assert torch.equal(res[0][1], res[1][1]), 'replicas diverged'
    assert float((res[0][1] - ref.arena.p).abs().max()) < 1e-6, 'DP on two half batches != full batch'
    assert torch.equal(res[0][2], res[1][2]) and float(res[1][2][0]) == 1.0      # running statistics broadcast from rank 0 once ...
    assert not torch.equal(res[0][3], res[1][3])                                  # ... and replica-local afterwards (no SyncBN)
    assert res[0][4] == [(128, 384), (0, 128)] and res[0][5] == 0.5               # top slice first, 1/world folded into SGD
    plain = FakeEngine(seed=1)                                   # the two halves together == the unsplit backward
    plain.lr = ref.lr
    segs, _ = plain.grad_buckets(1)
    for _ in range(3):
        plain.forward(X); plain.cross_entropy(8, Y); plain.backward(8, segments=segs); plain.adam_step()
    assert float((plain.arena.p - ref.arena.p).abs().max()) < 1e-6


def test_shard_indices_matches_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler
    for n, world in [(10, 4), (9843, 8), (7, 2)]:
        ds = list(range(n))
        for rank in range(world):
            ref = list(iter(DistributedSampler(ds, num_replicas=world, rank=rank, seed=0)))
            assert shard_indices(n, world, rank, seed=0) == ref
    assert shard_indices(6, 2, 1, shuffle=False) == [1, 3, 5]


def test_sliced_update_equals_the_single_update():
    """DataParallelTrainer(sliced_adam=True) (default) and sliced_adam=False walk the same parameters: one process, three steps."""
    torch.manual_seed(3)
    X = torch.randn(8, 8); Y = torch.randn(8, 8)
    out = []
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    dist.init_process_group('gloo', rank=0, world_size=1)        # (forced collectives need a group; one rank is enough here)
    try:
        for sliced in (True, False):
            eng = FakeEngine(seed=5)
            tr = DataParallelTrainer(eng, n_buckets=3, use_graphs=False, force_collectives=True, graph_collectives=False, sliced_adam=sliced)
            assert tr.sliced_adam == sliced and len(tr.slices) == 3
            for _ in range(3):
                tr.step_eager(X, Y)
            out.append(eng.arena.p.clone())
    finally:
        dist.destroy_process_group()
    assert torch.equal(out[0], out[1])


def test_reducer_is_a_noop_without_process_group():
    flat = torch.arange(8.0)
    red = BucketedGradReducer(flat, [(4, 8), (0, 4)])
    red.launch(0); red.launch(1); red.wait()
    assert torch.equal(flat, torch.arange(8.0))
    with pytest.raises(AssertionError):
        BucketedGradReducer(flat, [(5, 8), (0, 4)])
    broadcast_parameters(flat)                                   # no group -> no-op


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 6: the sharded optimizer (reduce-scatter -> Adam on the local shard -> all-gather overlapped with the next forward)
class FakeShardedEngine(FakeEngine):
    """FakeEngine + the phase interface ShardedDataParallelTrainer drives (VoxelEngine.forward_tokens / forward_blocks / forward_tail /
    loss_of_features / backward_begin / backward_segment / owning_grads / refresh_planes_range) and an optimizer WITH state (momentum), so
    that a shard updated from stale or foreign state would show."""

    def __init__(self, depth=6, width=8, seed=0):
        super().__init__(depth, width, seed)
        self.planes = self.arena.p.clone()                      # "hi / lo planes": what the forward reads (must be refreshed after a gather)
        self.plane_log = []
        self.update_log = []
        self.ws = type('WS', (), {})()
        self.ws.loss = torch.zeros(2)

    def W(self, i, flat=None):
        flat = self.planes if flat is None else flat             # the forward GEMMs read the planes, never the fp32 master
        return FakeEngine.W(self, i, flat)

    def refresh_weight_planes(self): self.planes.copy_(self.arena.p)
    def refresh_planes_range(self, s, e):
        self.planes[s:e] = self.arena.p[s:e]
        self.plane_log.append((s, e))
    def workspace(self, B): return self.ws
    def grad_buckets(self, n=3, blocks_per_bucket=None, block_counts=None):
        from simple3d_former_amd.engine import VoxelEngine
        return VoxelEngine.grad_buckets(self, n, blocks_per_bucket, block_counts)

    def forward_tokens(self, x):
        self.acts = [x]
        return self.ws

    def forward_blocks(self, ws, first, last):
        assert len(self.acts) == first + 1, 'forward ranges out of order'
        for i in range(first, last + 1):
            self.acts.append(torch.tanh(self.acts[-1] @ self.W(i).t()))

    def forward_tail(self, ws): return ws

    def loss_of_features(self, B, y, weight=None):
        self.ws.loss[0] = self.cross_entropy(B, y, weight)
        return self.ws.loss[0]

    def backward_begin(self, B):
        self._d = self.dout
        return self.ws

    def backward_segment(self, ws, first, last, with_tokenizer):
        d = self._d
        for i in range(first, last - 1, -1):
            d = d * (1 - self.acts[i + 1] ** 2)
            FakeEngine.W(self, i, self.arena.g).add_(d.t() @ self.acts[i])
            d = d @ self.W(i)
        self._d = d

    def backward(self, B, segments=None, on_segment=None):       # (the replicated trainer's eager path)
        self.backward_begin(B)
        for si, (first, last) in enumerate(segments):
            self.backward_segment(self.ws, first, last, si == len(segments) - 1)
            if on_segment:
                on_segment(si)

    def adam_apply(self, start, end, zero_grad=True, max_workgroups=0, wire=None):
        m = self.arena.m[start:end]
        m.mul_(0.9).add_(self.arena.g[start:end], alpha=self.grad_scale)
        self.arena.p[start:end].add_(m, alpha=-self.lr)
        self.planes[start:end] = self.arena.p[start:end]        # (the real kernel refreshes the planes of what it updates)
        if zero_grad:
            self.arena.g[start:end].zero_()
        self.update_log.append((start, end))

    def adam_step(self, zero_grad=True, wire=None):
        self.update_log = getattr(self, 'update_log', [])
        self.adam_apply(0, self.arena.numel, zero_grad)


def _sharded_worker(rank, world, port, q, sharded):
    _init_group(rank, world, port)
    try:
        torch.manual_seed(100)
        X = torch.randn(8, 8); Y = torch.randn(8, 8)
        eng = FakeShardedEngine(seed=rank)                       # different initial parameters per rank: the broadcast must fix it
        eng.lr = 0.05
        if sharded:
            tr = ShardedDataParallelTrainer(eng, bucket_blocks=[2, 2, 1, 1], use_graphs=False)
            assert tr.segments == [(5, 4), (3, 2), (1, 1), (0, 0)] and tr.fwd_ranges == [(0, 0), (1, 1), (2, 3), (4, 5)]
            assert tr.slices == [(256, 384), (128, 256), (64, 128), (0, 64)]
            assert tr.shards == [(s + rank * (e - s) // 2, s + (rank + 1) * (e - s) // 2) for s, e in tr.slices] and eng.grad_scale == 0.5
        else:
            tr = DataParallelTrainer(eng, n_buckets=3, use_graphs=False)
        sl = slice(rank * 4, rank * 4 + 4)
        losses = []
        for _ in range(4):
            losses.append(float(tr.step_eager(X[sl], Y[sl])))
        stale = None
        if sharded:
            # gather-pending: this rank holds the new values of its OWN shards only ...
            assert tr.pending
            stale = eng.arena.p.clone()
            sd = tr.state_dict if False else None
            tr.sync_parameters()                                 # ... until the gather phase has run
            assert not tr.pending
            # every step: Adam ran on this rank's shards only, in bucket order; the planes of every bucket were refreshed after its gather
            per_step = ['begin'] + [tuple(s_) for s_ in tr.shards]
            assert [u for u in eng.update_log if u != 'end'] == per_step * 4, eng.update_log
            assert eng.plane_log[:4] == [tuple(s_) for s_ in reversed(tr.slices)]
            assert torch.equal(eng.planes, eng.arena.p)
            # the optimizer state exists for the shards only: m outside them was never touched
            mask = torch.zeros(eng.arena.numel, dtype=torch.bool)
            for a, b in tr.shards:
                mask[a:b] = True
            assert not eng.arena.m[~mask].any() and eng.arena.m[mask].any()
            assert not eng.arena.g.any()                          # the whole gradient arena is zero again (own shard by Adam, the rest by the scatter)
        q.put(_to_wire((rank, eng.arena.p.clone(), losses, stale)))
    finally:
        dist.destroy_process_group()


def _run_two(target, *args):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_from_wire(q.get(timeout=120)) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_sharded_optimizer_two_ranks_bitwise_equal_the_replicated_trainer():
    """ShardedDataParallelTrainer on two ranks: replicas bitwise equal to each other AND to DataParallelTrainer's (all-reduce + replicated
    update) on the same two ranks -- same summed gradient, same update element by element -- and within fp32 rounding of one process on the
    full batch.  Before sync_parameters() a rank's copy of the OTHER rank's shards is one step stale."""
    sh = _run_two(_sharded_worker, True)
    rep = _run_two(_sharded_worker, False)
    assert torch.equal(sh[0][1], sh[1][1]), 'sharded replicas diverged'
    assert torch.equal(rep[0][1], rep[1][1]), 'replicated replicas diverged'
    assert torch.equal(sh[0][1], rep[0][1]), 'sharded update != replicated update'
    assert sh[0][2] != sh[1][2] and sh[0][2] == rep[0][2] and sh[1][2] == rep[1][2]       # per-rank losses of per-rank half batches, same in both designs
    assert not torch.equal(sh[0][3], sh[0][1]) and not torch.equal(sh[0][3], sh[1][3])   # gather-pending state differs per rank and from the result
    torch.manual_seed(100)
    X = torch.randn(8, 8); Y = torch.randn(8, 8)
    ref = FakeShardedEngine(seed=0)
    ref.lr = 0.05
    segs, _ = ref.grad_buckets(1)
    for _ in range(4):
        ref.forward_tokens(X); ref.forward_blocks(ref.ws, 0, 5); ref.cross_entropy(8, Y); ref.backward(8, segments=segs); ref.adam_step()
    assert float((sh[0][1] - ref.arena.p).abs().max()) < 1e-6, 'sharded DP on two half batches != one process on the full batch'


def test_sharded_trainer_one_rank_equals_the_plain_step():
    """world size 1, no process group: every collective is a no-op, a shard is the whole bucket -- the step IS forward / backward / update."""
    torch.manual_seed(7)
    X = torch.randn(8, 8); Y = torch.randn(8, 8)
    eng = FakeShardedEngine(seed=3)
    tr = ShardedDataParallelTrainer(eng, bucket_blocks=[3, 2, 1], use_graphs=False)
    assert tr.world == 1 and tr.shards == tr.slices and 'none' in tr.collectives_mode()
    ref = FakeShardedEngine(seed=3)
    segs, _ = ref.grad_buckets(1)
    for _ in range(3):
        tr.step_eager(X, Y)
        ref.forward_tokens(X); ref.forward_blocks(ref.ws, 0, 5); ref.cross_entropy(8, Y); ref.backward(8, segments=segs); ref.adam_step()
    tr.sync_parameters()
    assert torch.equal(eng.arena.p, ref.arena.p) and torch.equal(eng.arena.m, ref.arena.m)
    with pytest.raises(AssertionError):
        ShardedDataParallelTrainer(FakeShardedEngine(seed=1), bucket_blocks=[4, 1], use_graphs=False)      # counts must sum to the depth
