"""Data-parallel training of the voxel path (DataParallelTrainer) and of the point path (PointDataParallelTrainer): one
process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests), replicated
parameters, per-step gradient all-reduce.

Reference behaviour reproduced (train_cls_voxel.py): DDP constructor broadcast of rank-0 parameters (:155-159), mean
all-reduce of gradients overlapped with backward (:287), DistributedSampler's index partition without set_epoch
(:160-164).  MI355X-first differences: gradients live in ONE flat arena laid out in forward order, so each bucket is a
contiguous slice that completes back-to-front during backward -- no bucket copy-in/out; the 1/world averaging is folded
into the fused Adam kernel; backward is replayed as HIP-graph segments with the RCCL call of the previous bucket in
flight (RCCL's stream) while the next segment computes."""
import torch
import torch.distributed as dist


def shard_indices(n_samples, world_size, rank, seed=0, shuffle=True):
    """torch.utils.data.DistributedSampler's rule with epoch fixed at 0 (the reference never calls set_epoch):
    seeded permutation, padded by wrap-around to a multiple of world_size, then indices[rank::world_size]."""
    if shuffle:
        g = torch.Generator().manual_seed(seed)
        idx = torch.randperm(n_samples, generator=g).tolist()
    else:
        idx = list(range(n_samples))
    total = (n_samples + world_size - 1) // world_size * world_size
    pad = total - len(idx)
    if pad:
        idx += (idx * ((pad + len(idx) - 1) // len(idx)))[:pad]
    return idx[rank:total:world_size]


class BucketedGradReducer:
    """Sum-all-reduce of contiguous slices of one flat gradient tensor, launched asynchronously bucket by bucket.
    prepare(start, end), if given, runs on the current stream right before bucket [start, end) is handed to the collective (the
    bf16 wire format packs the finished fp32 slice into the flat bf16 tensor there)."""

    def __init__(self, flat_grad, slices, group=None, force=False, prepare=None, standin_gbps=0.0, standin_latency_us=0.0):
        """standin_gbps > 0 (diagnostics, one rank): instead of the collective -- which launches NOTHING at world size 1 -- bucket i is
        copied into a scratch buffer by a few workgroups paced to about that rate, on a side stream forked from the current one: a kernel of
        the duration and footprint a ring all-reduce of the bucket would have, so that the cost of a LIVE side branch in a captured step
        graph (profiles/r03_graph_branch_probe.txt) can be measured without peers (profiles/r05_dp_branch_tax.txt)."""
        self.flat, self.slices, self.group = flat_grad, list(slices), group
        self.force = force                   # issue the collective even at world size 1 (exercises the RCCL path)
        self.prepare = prepare
        self.skip = False                    # diagnostics only (bench.py): time the step with the collectives suppressed
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._works = {}
        self.standin_gbps = float(standin_gbps)
        self.standin_latency_us = float(standin_latency_us)             # fixed cost per collective added to the stand-in (ring start-up)
        self._standin_stream = None
        self._standin_buf = None
        covered = sorted(self.slices)
        assert covered[0][0] == 0 and covered[-1][1] == flat_grad.numel() and \
            all(a[1] == b[0] for a, b in zip(covered, covered[1:])), 'buckets must tile the arena exactly'

    def launch(self, i):
        s, e = self.slices[i]
        if self.prepare is not None:
            self.prepare(s, e)
        if (self.world == 1 and not self.force) or self.skip:
            return
        if self.standin_gbps > 0 and self.world == 1 and self.flat.is_cuda:
            self._works[i] = self._standin(s, e)
            return
        self._works[i] = dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _standin(self, s, e):
        import ctypes
        from . import _lib as L
        if self._standin_stream is None:
            self._standin_stream = torch.cuda.Stream()
            self._standin_buf = torch.empty_like(self.flat)
        side, cur = self._standin_stream, torch.cuda.current_stream()
        side.wait_stream(cur)                                           # fork: the bucket is final on the compute stream
        nbytes = (e - s) * self.flat.element_size()
        # duration = latency + bytes / rate, expressed as one rate for the paced copy
        gbps = self.standin_gbps
        if self.standin_latency_us > 0:
            gbps = nbytes / (self.standin_latency_us * 1e3 + nbytes / self.standin_gbps)
        with torch.cuda.stream(side):
            L.check(L.lib().s3d_debug_paced_copy(ctypes.c_void_p(self._standin_buf.data_ptr() + s * self.flat.element_size()),
                                                 ctypes.c_void_p(self.flat.data_ptr() + s * self.flat.element_size()),
                                                 ctypes.c_long(nbytes), ctypes.c_float(gbps), L.current_stream()), 'paced_copy')
        return side

    def wait_one(self, i):
        """The current stream waits for bucket i's collective (no host block on GPU backends); other buckets stay in flight."""
        w = self._works.pop(i, None)
        if w is None:
            return
        if isinstance(w, torch.cuda.Stream):
            torch.cuda.current_stream().wait_stream(w)                  # join of the stand-in branch
        else:
            w.wait()

    def wait(self):
        for i in sorted(self._works):
            self.wait_one(i)


def broadcast_parameters(flat_param, src=0, group=None):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_param, src=src, group=group)


def graph_collectives_preflight(device_index, group=None, timeout_s=None, sharded=False):
    """Can this job replay RCCL all-reduces captured in a HIP graph?  Every rank starts `_graph_collective_preflight.py` as a child
    process on its own device (same RANK / WORLD_SIZE, rendezvous on a free port that rank 0 picks and broadcasts), waits at most timeout_s
    (S3D_PREFLIGHT_TIMEOUT, default 150 s) and kills it otherwise; the verdicts are combined with a MIN all-reduce over the caller's
    (eager) process group, so all ranks take the same branch.  Returns (ok, detail).  ~10 - 20 s, once per trainer."""
    import os
    import subprocess
    import sys
    timeout_s = float(os.environ.get('S3D_PREFLIGHT_TIMEOUT', '150')) if timeout_s is None else timeout_s
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    env = dict(os.environ)
    # rendezvous of the children: rank 0 picks a free port on its own address and tells the others over the caller's (eager) process group --
    # works for tcp:// / file:// initialised jobs without MASTER_* variables, on several nodes, and for two trainers on one host
    addr, port = env.get('MASTER_ADDR', '127.0.0.1'), int(env.get('MASTER_PORT', '29500')) + 13
    if dist.is_initialized() and world > 1:
        import socket
        if rank == 0:
            with socket.socket() as sk:
                sk.bind(('', 0))
                port = sk.getsockname()[1]
            if 'MASTER_ADDR' not in env:
                try:
                    addr = socket.gethostbyname(socket.gethostname())
                except OSError:
                    addr = '127.0.0.1'
        box = [addr, port]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        addr, port = box[0], int(box[1])
    env.update(RANK=str(rank), WORLD_SIZE=str(world), S3D_PREFLIGHT_DEVICE=str(device_index), MASTER_ADDR=str(addr), MASTER_PORT=str(port),
               S3D_PREFLIGHT_SHARDED='1' if sharded else '0')
    env.pop('TORCHELASTIC_RUN_ID', None)
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_graph_collective_preflight.py')
    detail = ''
    try:
        proc = subprocess.Popen([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        try:
            out, _ = proc.communicate(timeout=timeout_s)
            ok = proc.returncode == 0 and b'S3D_PREFLIGHT_OK' in out
            if not ok:
                detail = f'child exit code {proc.returncode}: ' + out.decode(errors='replace').strip().splitlines()[-1][-200:] if out.strip() else f'child exit code {proc.returncode}'
        except subprocess.TimeoutExpired:
            proc.kill()
            proc.communicate()
            ok, detail = False, f'no answer within {timeout_s:.0f} s (child killed)'
    except OSError as e:
        ok, detail = False, f'could not start the child: {e}'
    if dist.is_initialized() and world > 1:
        dev = torch.device('cuda', device_index) if dist.get_backend(group) == 'nccl' else torch.device('cpu')
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if ok and int(flag.item()) == 0:
            detail = 'another rank failed the preflight'
        ok = bool(int(flag.item()))
    if not ok:
        import warnings
        warnings.warn(f'[s3d] graph-collective preflight failed ({detail}): one HIP graph per backward segment with host-launched collectives instead')
    return ok, detail


class DataParallelTrainer:
    """Fused training step (forward, loss, backward, all-reduce, Adam) for one rank."""

    def __init__(self, engine, n_buckets=None, group=None, use_graphs=True, force_collectives=False, graph_collectives='auto',
                 wire='fp32', event_graph=False, sliced_adam=True, standin_gbps=0.0, blocks_per_bucket=None, standin_latency_us=0.0):
        """event_graph (opt-in, with use_graphs): forward + the whole backward are captured as ONE flat graph with an external
        event-record node behind every backward segment (s3d_graph_marker / s3d_graph_events_at_markers); the all-reduce of bucket k is
        launched from the host on a side stream that waits for "segment k done" only, so it overlaps the later segments and the
        compute stream never sees a graph boundary or a c10d event packet.  No collective lives inside a graph (nothing that could
        hang with real peers).  Built and measured in round 3 (tools/dp_overhead_probe.py, profiles/r03_dp_overhead_probe.txt): on
        this runtime an event-record node inside a graph costs ~20 us -- MORE than the graph boundary it replaces (~10 us) -- so the
        default stays event_graph=False: one graph per segment with the collectives launched in between.
        wire: 'fp32' all-reduces the gradient arena itself (DDP's arithmetic); 'bf16' rounds every finished bucket to bf16
        (s3d_pack_bf16, on the compute stream right after its backward segment), all-reduces HALF the bytes over xGMI and lets the
        Adam kernel read the bf16 sum (s3d_adam_step_wire) -- gradient compression as in DDP's bf16_compress_hook.
        graph_collectives: capture the WHOLE step -- backward segments, the all-reduce of every bucket (RCCL calls are
        capturable: tools/probes/rccl_graph_probe.py), Adam -- into ONE HIP graph instead of one graph per segment with the
        collectives launched from the host in between: the compute stream then never leaves the graph (leaving it costs ~0.11 ms of a
        1.7 ms cfg-2 step, every further boundary ~0.02 ms: profiles/r03_dp_bucket_sweep.txt).  'auto' (default): when there are
        collectives to issue over RCCL, a throw-away child process group first proves on the job's own ranks and devices that captured
        all-reduces replay correctly (graph_collectives_preflight); only then is the step captured that way -- and still replayed once
        under a watchdog -- otherwise the trainer falls back to one graph per segment.  True: no preflight; False: never.
        Defaults (round 4): fp32 on the wire = DDP's own arithmetic (bf16 stays the flagged option); n_buckets = None picks 4 buckets
        (blocks 11-6 | 5-3 | 2-1 | 0 + tokenizer) when the collectives are graph nodes -- extra buckets are free there (1.710 ms with 4
        vs 1.714 with 2 at one rank) and the exposed last bucket shrinks from 43 MB to 7 MB -- and 2 when they are host-launched
        (~0.02 ms per extra graph boundary): profiles/r04_dp_sweep.txt."""
        self.eng = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.preflight = None               # (ok, detail) of the 'auto' decision, for bench.py's line
        if graph_collectives == 'auto':
            rccl = dist.is_initialized() and dist.get_backend(group) == 'nccl' and engine.device.type == 'cuda'
            if use_graphs and rccl and (self.world > 1 or force_collectives) and not event_graph:
                self.preflight = graph_collectives_preflight(engine.device.index if engine.device.index is not None else torch.cuda.current_device(), group)
                graph_collectives = self.preflight[0]
            else:
                graph_collectives = False
        self.graph_collectives = bool(graph_collectives)
        n_buckets_given = n_buckets
        if n_buckets is None:
            n_buckets = 4 if (self.graph_collectives and use_graphs) else 2
        broadcast_parameters(engine.arena.p, 0, group)                 # DDP-constructor broadcast (C2)
        engine.refresh_weight_planes()
        live = self.world > 1 or force_collectives
        if blocks_per_bucket is None and n_buckets_given is None and live and getattr(engine, 'depth', 0) >= 8:
            # round 5 default: UNIFORM buckets of depth / 4 blocks (cfg-2: 11-9 | 8-6 | 5-3 | 2-0 + tokenizer).  With a live kernel of the
            # duration of an 8-rank fp32 ring all-reduce per bucket on the side branch (profiles/r05_dp_branch_tax.txt) the step takes
            # 1.93 ms against 2.06 ms with the geometric buckets of rounds 3 - 4 (the wire starts after three blocks of backward instead of
            # six, and the whole gradient's wire time -- 0.7 ms -- is as long as the backward); 6 / 12 buckets: 2.11 / 2.38 ms.
            blocks_per_bucket = max(1, engine.depth // 4)
        if blocks_per_bucket and live:
            self.segments, self.slices = engine.grad_buckets(blocks_per_bucket=blocks_per_bucket)
        else:
            self.segments, self.slices = engine.grad_buckets(n_buckets if live else 1)
        if wire not in ('fp32', 'bf16'):
            raise ValueError(f"wire must be 'fp32' or 'bf16', not {wire!r}")
        self.wire = None
        if wire == 'bf16':
            self.wire = torch.zeros(engine.arena.g.numel(), dtype=torch.bfloat16, device=engine.arena.g.device)
            self.reducer = BucketedGradReducer(self.wire, self.slices, group, force=force_collectives,
                                               prepare=lambda s, e: engine.pack_grads(s, e, self.wire), standin_gbps=standin_gbps,
                                               standin_latency_us=standin_latency_us)
        else:
            self.reducer = BucketedGradReducer(engine.arena.g, self.slices, group, force=force_collectives, standin_gbps=standin_gbps,
                                               standin_latency_us=standin_latency_us)
        # keep the lr / betas / eps the engine was built with (the reference's default lr is 0.05, train_cls_voxel.py:373); only the
        # 1/world averaging is this trainer's business
        engine.set_optimizer(grad_scale=1.0 / self.world)
        if self.world > 1:                   # DDP ranks draw different dropout masks (independent RNG streams per process)
            engine.dropout_seed.add_(1000003 * dist.get_rank(group))
        self.use_graphs = use_graphs
        self.event_graph = bool(event_graph) and not self.graph_collectives
        self._side = None                   # the stream the bucket collectives are issued from in event_graph mode
        self._cap = None
        # optimizer.step() bucket by bucket (round 5): Adam on bucket k's arena slice as soon as ITS all-reduce has finished, while the
        # later buckets are still on the wire -- only the last (smallest) bucket's wire time and update stay exposed, instead of every
        # bucket's wait followed by one update over the whole arena.  Same arithmetic (s3d_adam_begin + s3d_adam_apply == s3d_adam_step).
        self.sliced_adam = sliced_adam and hasattr(engine, 'adam_apply') and len(self.slices) > 1

    def collectives_mode(self):
        """How this trainer's captured step issues the bucket all-reduces (bench.py prints it)."""
        if self.world == 1 and not self.reducer.force:
            return 'none'
        if not self.use_graphs:
            return 'host-launched between eager backward segments'
        if self.graph_collectives:
            return 'captured in the step graph'
        return 'host-launched on graph events' if self.event_graph else 'host-launched between graph segments'

    def set_optimizer(self, lr=None, betas=None, eps=None):
        self.eng.set_optimizer(lr=lr, betas=betas, eps=eps, grad_scale=1.0 / self.world)

    def _update(self, k=None):
        """optimizer.step(): k = None -> every bucket (waits for the collectives as it goes); k -> bucket k only (sliced mode)."""
        eng = self.eng
        if not self.sliced_adam:
            self.reducer.wait()
            eng.adam_step(zero_grad=True, wire=self.wire)
            return
        ks = range(len(self.slices)) if k is None else [k]
        for i in ks:
            if i == 0:
                eng.adam_begin()                                        # step count / bias corrections once per step
            self.reducer.wait_one(i)
            eng.adam_apply(*self.slices[i], zero_grad=True, wire=self.wire)
            if i == len(self.slices) - 1:
                eng.adam_end()

    # ---- eager step -------------------------------------------------------------------------------------------
    def step_eager(self, x, y, weight=None):
        eng, B = self.eng, x.shape[0]
        eng.advance_dropout_seed()
        loss = eng.forward_loss(x, y, weight)
        with eng.owning_grads():            # this step owns the gradient arena (zeroed by its Adam, one backward): the grouped wgrads may store
            eng.backward(B, segments=self.segments, on_segment=self.reducer.launch)
        self._update()
        return loss

    # ---- HIP-graph step ---------------------------------------------------------------------------------------
    def _phase(self, k, B, sx, sy, weight):
        eng = self.eng
        if k == 0:
            eng.advance_dropout_seed()                                  # captured: every replay draws fresh masks
            eng.forward_loss(sx, sy, weight)
            ws = eng.backward_begin(B)
        else:
            ws = eng.workspace(B)
        first, last = self.segments[k]
        with eng.owning_grads():            # scoped to the trainer's own backward: a direct engine.backward() elsewhere keeps accumulating
            eng.backward_segment(ws, first, last, k == len(self.segments) - 1)

    def capture(self, B, weight=None):
        eng = self.eng
        sx = torch.zeros(B, 1, eng.V, eng.V, eng.V, dtype=torch.float32, device=eng.device)
        sy = torch.zeros(B, dtype=torch.int64, device=eng.device)
        state = (eng.arena.p, eng.arena.m, eng.arena.v, eng.arena.g, eng.adam_state, eng.dropout_seed)
        snap = [t.clone() for t in state]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                                   # warm-up: kernel attributes, workspaces
            for k in range(len(self.segments)):
                self._phase(k, B, sx, sy, weight)
            self._update()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for t, sv in zip(state, snap):
            t.copy_(sv)
        eng.refresh_weight_planes()
        torch.cuda.synchronize()
        if self.graph_collectives:
            self.reducer.launch(0); self.reducer.wait()                # communicator set-up outside the capture
            eng.arena.g.zero_()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for k in range(len(self.segments)):
                    self._phase(k, B, sx, sy, weight)
                    self.reducer.launch(k)                              # a side branch of the graph: overlaps the next segment
                self._update()
            self._cap = dict(B=B, graphs=[], whole=g, x=sx, y=sy, loss=eng.workspace(B).loss, epoch=eng.capture_epoch)
            self._guarded_first_replay(g, state)
            return self._cap
        self._release_events()
        if self.collectives_mode() == 'none' and self.wire is None:
            # one replica, nothing to reduce: the engine's own step (with its update-beside-backward schedule) as ONE graph
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                eng.train_step(sx, sy, weight)
            self._cap = dict(B=B, graphs=[g], opt=None, x=sx, y=sy, loss=eng.workspace(B).loss, epoch=eng.capture_epoch)
            return self._cap
        if self.event_graph:
            import ctypes
            from . import _lib as L
            lib = L.lib()
            events = []
            for _ in self.segments:
                ev = ctypes.c_void_p()
                L.check(lib.s3d_event_create(ctypes.byref(ev)), 'event_create')
                events.append(ev)
            g = torch.cuda.CUDAGraph(keep_graph=True)                   # instantiated at its first replay, i.e. after the edit below
            with torch.cuda.graph(g):
                for k in range(len(self.segments)):
                    self._phase(k, B, sx, sy, weight)
                    L.check(lib.s3d_graph_marker(k, L.current_stream()), 'graph_marker')
            L.check(lib.s3d_graph_events_at_markers(ctypes.c_void_p(g.raw_cuda_graph()), (ctypes.c_void_p * len(events))(*events),
                                                    len(events)), 'graph_events_at_markers')
            g_opt = self._capture_update()
            if self._side is None:
                self._side = torch.cuda.Stream()
            self._cap = dict(B=B, graphs=[], fwd_bwd=g, events=events, opt=g_opt, x=sx, y=sy, loss=eng.workspace(B).loss,
                             epoch=eng.capture_epoch)
            return self._cap
        graphs = []
        for k in range(len(self.segments)):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._phase(k, B, sx, sy, weight)
            graphs.append(g)
        g_opt = self._capture_update()
        self._cap = dict(B=B, graphs=graphs, opt=g_opt, x=sx, y=sy, loss=eng.workspace(B).loss, epoch=eng.capture_epoch)
        return self._cap

    def _capture_update(self):
        """The optimizer as graph(s) for the host-launched modes: one graph, or (sliced) one per bucket -- replayed behind that bucket's wait.
        The collectives are NOT part of these graphs (the reducer has nothing in flight while they are captured)."""
        eng = self.eng
        if not self.sliced_adam:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                eng.adam_step(zero_grad=True, wire=self.wire)
            return [g]
        graphs = []
        for k in range(len(self.slices)):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._update(k)
            graphs.append(g)
        return graphs

    def _replay_update(self, graphs):
        if len(graphs) == 1 and not self.sliced_adam:
            self.reducer.wait()
            graphs[0].replay()
            return
        for k, g in enumerate(graphs):
            self.reducer.wait_one(k)
            g.replay()

    def _release_events(self):
        if self._cap is not None and self._cap.get('events'):
            from . import _lib as L
            torch.cuda.synchronize()
            for ev in self._cap['events']:
                L.lib().s3d_event_destroy(ev)
            self._cap['events'] = []

    def __del__(self):
        try:
            self._release_events()
        except Exception:                   # interpreter shutdown
            pass

    def _guarded_first_replay(self, graph, state, timeout_s=None):
        """First replay of a graph that holds RCCL kernels, under a watchdog: a capture problem that only shows with real peers
        would otherwise hang the job forever inside the device queue.  The replay is enqueued, the host polls an event for at
        most timeout_s (S3D_GRAPH_COLLECTIVE_TIMEOUT, default 120 s) and, if it never completes, reports and leaves the process
        with exit code 3 (a hung device queue cannot be recovered in-process).  The training state is restored afterwards."""
        import os
        import sys
        import time
        timeout_s = float(os.environ.get('S3D_GRAPH_COLLECTIVE_TIMEOUT', '120')) if timeout_s is None else timeout_s
        snap = [t.clone() for t in state]
        torch.cuda.synchronize()
        graph.replay()
        ev = torch.cuda.Event()
        ev.record()
        t0 = time.monotonic()
        while not ev.query():
            if time.monotonic() - t0 > timeout_s:
                rank = dist.get_rank(self.group) if dist.is_initialized() else 0
                sys.stderr.write(f'[s3d] rank {rank}: the first replay of the step graph with captured all-reduces did not finish within '
                                 f'{timeout_s:.0f} s -- giving up (use the host-launched collectives, the default)\n')
                sys.stderr.flush()
                os._exit(3)
            time.sleep(0.002)
        for t, sv in zip(state, snap):
            t.copy_(sv)
        self.eng.refresh_weight_planes()
        torch.cuda.synchronize()
        self.first_replay_s = time.monotonic() - t0

    def step_graph(self):
        """Replays the captured step on the static buffers (cap['x'], cap['y'] must already hold the batch)."""
        cap = self._cap
        if cap['epoch'] != self.eng.capture_epoch:
            raise RuntimeError('the captured step is stale (set_dropout changed a value baked into the graphs): capture() again')
        if 'whole' in cap:
            cap['whole'].replay()
            return cap['loss'][0]
        if cap.get('fwd_bwd') is not None:
            import ctypes
            from . import _lib as L
            lib, cur, side = L.lib(), torch.cuda.current_stream(), self._side
            cap['fwd_bwd'].replay()
            for k, ev in enumerate(cap['events']):
                # the side stream waits for the event behind segment k -- recorded inside the graph that is already running -- and the
                # collective (and the bf16 pack in front of it) is issued from there: c10d synchronises with the SIDE stream
                L.check(lib.s3d_stream_wait_event(ctypes.c_void_p(side.cuda_stream), ev), 'stream_wait_event')
                with torch.cuda.stream(side):
                    self.reducer.launch(k)
            cur.wait_stream(side)           # whatever the side stream did besides the collectives (wire packing at world size 1)
            self._replay_update(cap['opt'])                             # the compute stream waits for the collectives bucket by bucket
            return cap['loss'][0]
        for k, g in enumerate(cap['graphs']):
            g.replay()
            self.reducer.launch(k)          # RCCL all-reduce of bucket k overlaps the next segment's replay
        if cap['opt'] is not None:
            self._replay_update(cap['opt'])
        else:
            self.reducer.wait()
        return cap['loss'][0]

    def step(self, x, y, weight=None):
        if not self.use_graphs:
            return self.step_eager(x, y, weight)
        if self._cap is None or self._cap['B'] != x.shape[0] or self._cap['epoch'] != self.eng.capture_epoch:
            self.capture(x.shape[0], weight)
        self._cap['x'].copy_(x, non_blocking=True)
        self._cap['y'].copy_(y, non_blocking=True)
        return self.step_graph()


class ShardedDataParallelTrainer:
    """Data-parallel step with a SHARDED optimizer (round 6; VERDICT r05 items 1 - 2 of "what's missing"):

        backward segment k done  ->  reduce-scatter of bucket k (every rank keeps the SUM of 1 / world of the bucket's gradient, in place)
                                 ->  Adam on that shard only (m, v and the update exist for 1 / world of the parameters per rank)
        next step, before range i of the forward  ->  all-gather of the bucket range i reads (fp32 parameters, in place in the arena)
                                 ->  hi / lo weight planes of the bucket (s3d_split_bf16 on the gathered slice)

    Same wire bytes as the all-reduce it replaces (reduce-scatter + all-gather ARE a ring all-reduce's two halves), but (a) the replicated
    127 us Adam of cfg-2 becomes 1 / world of it per rank, (b) the all-gather half overlaps the NEXT forward instead of sitting between the
    last backward segment and the optimizer, bucket by bucket in the order the forward needs them, so the exposed wire time is one small
    reduce-scatter + one small all-gather (the last bucket: block 0 + tokenizer) instead of the tail of a whole ring.  DDP's own schedule
    (train_cls_voxel.py:155-159,287-288: all-reduce overlapped with backward, then optimizer.step()) is what DataParallelTrainer mirrors.

    Arithmetic: identical to the replicated path -- the same summed gradient, the same Adam element by element (each element is updated by
    exactly one rank and copied to the others), the same rne split -- so replicas stay bitwise equal to each other by construction; against
    DataParallelTrainer the parameters are bitwise equal wherever reduce-scatter and all-reduce add in the same order (always at two ranks).

    The step leaves the parameters GATHER-PENDING: every rank holds the new values of its own shards only until the next step's gather phase
    (or sync_parameters()) has run; state_dict() / evaluation go through sync_parameters().

    emulate_world = E with standin_gbps > 0 (one-rank diagnostics, profiles/r06_dp_sharded.txt): the collectives are replaced by copy kernels
    paced to an E-rank ring's reduce-scatter / all-gather duration on the side stream and Adam runs on 1 / E of every bucket -- the timing of an
    E-rank step on one GPU; the parameters of such a run mean nothing."""

    def __init__(self, engine, group=None, bucket_blocks=None, use_graphs=True, graph_collectives='auto', force_collectives=False,
                 standin_gbps=0.0, standin_latency_us=0.0, emulate_world=0):
        self.eng, self.group = engine, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.cuda = bool(engine.arena.g.is_cuda)
        self.emulate_world = int(emulate_world) if (emulate_world and self.world == 1 and standin_gbps > 0) else 0
        self.standin_gbps, self.standin_latency_us = float(standin_gbps), float(standin_latency_us)
        self.force = bool(force_collectives)
        self.preflight = None
        live = self.world > 1 or self.force
        if graph_collectives == 'auto':
            rccl = dist.is_initialized() and self.cuda and dist.get_backend(group) == 'nccl'
            if use_graphs and rccl and live:
                self.preflight = graph_collectives_preflight(engine.device.index if engine.device.index is not None else torch.cuda.current_device(), group,
                                                             sharded=True)
                graph_collectives = self.preflight[0]
            else:
                graph_collectives = self.world == 1          # nothing (or only stand-in kernels) on the side stream: always capturable
        self.graph_collectives = bool(graph_collectives)
        self.use_graphs = bool(use_graphs) and self.cuda
        broadcast_parameters(engine.arena.p, 0, group)                 # DDP-constructor broadcast (train_cls_voxel.py:155-159)
        engine.refresh_weight_planes()
        depth = getattr(engine, 'depth', 0)
        if bucket_blocks is None:
            # backward order; the LAST bucket (block 0 + tokenizer / positional embedding) is the only one whose collectives are exposed
            # depth 12: three buckets of four blocks -- measured at one rank with stand-in collectives of an 8-rank ring's duration
            # (profiles/r06_dp_sharded.txt): 4,4,4 1.93 ms, 4,4,3,1 1.96, 3,3,3,2,1 2.08, 6,5,1 2.04 (every further bucket costs ~50 us of
            # fork / join whatever it hides; a smaller last bucket exposes less wire but does not pay for its own branch)
            bucket_blocks = [4, 4, 4] if depth == 12 else ([depth - depth // 2, depth // 2] if depth >= 2 else [depth])
        self.segments, self.slices = engine.grad_buckets(block_counts=bucket_blocks)
        n = len(self.slices)
        self.fwd_ranges = [(last, first) for (first, last) in reversed(self.segments)]        # ascending; range i reads bucket n - 1 - i
        W = self.emulate_world or self.world
        self.shards = []
        for (s0, e0) in self.slices:
            assert (e0 - s0) % (8 * W) == 0, (f'bucket [{s0}, {e0}) is not divisible into {W} shards of whole 8-element groups: the arena '
                                             f'aligns block starts and its end to engine.BUCKET_ALIGN = 512')
            Lk = (e0 - s0) // W
            r = 0 if self.emulate_world else self.rank
            self.shards.append((s0 + r * Lk, s0 + (r + 1) * Lk))
        engine.set_optimizer(grad_scale=1.0 / self.world)
        if self.world > 1:
            engine.dropout_seed.add_(1000003 * self.rank)              # DDP ranks draw different dropout masks
        self._side = torch.cuda.Stream() if self.cuda else None
        self._standin_buf = None
        self._cap = None
        self.pending = False            # True: shards updated, gather phase not yet run
        self._rs_native = True          # reduce_scatter_tensor available on this backend (gloo: emulated by all-reduce + slice)
        self._gloo = dist.is_initialized() and dist.get_backend(group) == 'gloo'

    # ---- the two collectives (or their stand-ins), on the current stream ------------------------------------------------------------
    def _standin(self, s0, e0, elem):
        import ctypes
        from . import _lib as L
        flat = self.eng.arena.g
        if self._standin_buf is None:
            self._standin_buf = torch.empty_like(flat)
        E = self.emulate_world
        nbytes = (e0 - s0) * elem
        wire = nbytes * (E - 1) / E                                    # what one rank sends (and receives) in a ring reduce-scatter / all-gather
        dur_ns = self.standin_latency_us * 1e3 + wire / self.standin_gbps      # GB/s == bytes / ns
        gbps = nbytes / dur_ns
        L.check(L.lib().s3d_debug_paced_copy(ctypes.c_void_p(self._standin_buf.data_ptr() + s0 * 4), ctypes.c_void_p(flat.data_ptr() + s0 * 4),
                                             ctypes.c_long(nbytes), ctypes.c_float(gbps), L.current_stream()), 'paced_copy')

    def _reduce_scatter(self, k):
        """bucket k of the gradient arena -> this rank's shard holds the sum over ranks; the rest of the bucket is zeroed (gradients that
        ACCUMULATE -- LayerNorm, embeddings, head -- start the next backward from zero everywhere, as after the replicated Adam)."""
        (s0, e0), (a, b) = self.slices[k], self.shards[k]
        g = self.eng.arena.g
        if self.emulate_world:
            self._standin(s0, e0, 4)
        elif self.world > 1 or self.force:
            done = False
            if self._rs_native and not self._gloo:
                dist.reduce_scatter_tensor(g[a:b], g[s0:e0], op=dist.ReduceOp.SUM, group=self.group)      # in place: output = own chunk of the input
                done = True
            if not done:                                               # gloo has no reduce-scatter: all-reduce, keep the own slice (same sums)
                dist.all_reduce(g[s0:e0], op=dist.ReduceOp.SUM, group=self.group)
        if a > s0:
            g[s0:a].zero_()
        if e0 > b:
            g[b:e0].zero_()

    def _all_gather(self, k):
        """fp32 parameters of bucket k: every rank's shard -> every rank (in place), then the bucket's hi / lo weight planes."""
        (s0, e0), (a, b) = self.slices[k], self.shards[k]
        p = self.eng.arena.p
        if self.emulate_world:
            self._standin(s0, e0, 4)
        elif self.world > 1 or self.force:
            if self._gloo:
                # (the CPU / one-GPU tests: gloo has no all_gather_into_tensor for device tensors -- one broadcast per shard, same result)
                Lk = (e0 - s0) // self.world
                for r in range(self.world):
                    dist.broadcast(p[s0 + r * Lk:s0 + (r + 1) * Lk], src=dist.get_global_rank(self.group, r) if self.group is not None else r,
                                   group=self.group)
            else:
                dist.all_gather_into_tensor(p[s0:e0], p[a:b], group=self.group)      # RCCL gathers in place (input = own chunk of the output)
        self.eng.refresh_planes_range(s0, e0)

    # ---- stream plumbing (no-ops on the CPU stand-in engine of the gloo tests) --------------------------------------------------------
    def _on_side(self, fn):
        if not self.cuda:
            return fn()
        with torch.cuda.stream(self._side):
            return fn()

    def _side_waits_main(self):
        if self.cuda:
            self._side.wait_stream(torch.cuda.current_stream())

    def _main_waits_side(self):
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self._side)

    def _event_on_side(self):
        if not self.cuda:
            return None
        ev = torch.cuda.Event()
        ev.record(self._side)
        return ev

    def _main_waits(self, ev):
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    # ---- the step ----------------------------------------------------------------------------------------------------------------------
    def _gather_phase(self):
        """Side stream: all-gather + plane refresh of every bucket in the order the forward needs them (last bucket first); returns the
        per-bucket events the forward ranges wait for."""
        n = len(self.slices)
        ready = [None] * n
        self._side_waits_main()
        for k in reversed(range(n)):
            self._on_side(lambda k=k: self._all_gather(k))
            if k == n - 1:
                self._on_side(self.eng.adam_end)                       # padded tokenizer-weight planes (whole weight: the last bucket)
            ready[k] = self._event_on_side()
        self.pending = False
        return ready

    def _step_body(self, x, y, weight, compute):
        """compute(i, fn): runs (eager / whole-step capture) or replays (segment graphs) compute phase i on the main stream."""
        eng, n = self.eng, len(self.slices)
        B = x.shape[0]
        ready = self._gather_phase()
        state = {}

        def phase_tokens():
            eng.advance_dropout_seed()
            state['ws'] = eng.forward_tokens(x)
            eng.forward_blocks(state['ws'], *self.fwd_ranges[0])
        self._main_waits(ready[n - 1])
        compute(0, phase_tokens)
        for i in range(1, n):
            self._main_waits(ready[n - 1 - i])
            compute(i, lambda i=i: eng.forward_blocks(state.get('ws') or eng.workspace(B), *self.fwd_ranges[i]))

        def phase_loss_and_first_segment():
            ws = state.get('ws') or eng.workspace(B)
            eng.forward_tail(ws)
            state['loss'] = eng.loss_of_features(B, y, weight)
            ws = eng.backward_begin(B)
            with eng.owning_grads():
                eng.backward_segment(ws, *self.segments[0], n == 1)

        def phase_segment(k):
            with eng.owning_grads():
                eng.backward_segment(eng.workspace(B), *self.segments[k], k == n - 1)
        for k in range(n):
            compute(n + k, phase_loss_and_first_segment if k == 0 else (lambda k=k: phase_segment(k)))
            self._side_waits_main()                                    # bucket k's gradients are final on the main stream

            def comm(k=k):
                self._reduce_scatter(k)
                if k == 0:
                    eng.adam_begin()                                   # step count / bias corrections once per step
                eng.adam_apply(*self.shards[k], zero_grad=True)
            self._on_side(comm)
        self._main_waits_side()
        self.pending = True
        return state.get('loss')

    def step_eager(self, x, y, weight=None):
        return self._step_body(x, y, weight, lambda i, fn: fn())

    def sync_parameters(self):
        """Completes a step: gathers the parameters the last step's shards updated (a no-op when nothing is pending)."""
        if self.pending:
            self._gather_phase()
            self._main_waits_side()

    def state_dict(self):
        self.sync_parameters()
        return self.eng.state_dict()

    def collectives_mode(self):
        if self.world == 1 and not self.force and not self.emulate_world:
            return 'none (one rank: sharded update == sliced update)'
        kind = 'stand-in copy kernels' if self.emulate_world else 'reduce-scatter + all-gather'
        if not self.use_graphs:
            return f'{kind}, host-launched between eager phases'
        return f'{kind}, ' + ('captured in the step graph' if self.graph_collectives else 'host-launched between graph phases')

    # ---- HIP-graph step ------------------------------------------------------------------------------------------------------------------
    def capture(self, B, weight=None):
        eng = self.eng
        sx = torch.zeros(B, 1, eng.V, eng.V, eng.V, dtype=torch.float32, device=eng.device)
        sy = torch.zeros(B, dtype=torch.int64, device=eng.device)
        state = (eng.arena.p, eng.arena.m, eng.arena.v, eng.arena.g, eng.adam_state, eng.dropout_seed)
        snap = [t.clone() for t in state]
        warm = torch.cuda.Stream()
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):                                   # warm-up: kernel attributes, workspaces, communicator set-up
            self._step_body(sx, sy, weight, lambda i, fn: fn())
            self.sync_parameters()
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        for t, sv in zip(state, snap):
            t.copy_(sv)
        eng.refresh_weight_planes()
        self.pending = False
        torch.cuda.synchronize()
        if self.graph_collectives:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._step_body(sx, sy, weight, lambda i, fn: fn())
            self._cap = dict(B=B, whole=g, x=sx, y=sy, loss=eng.workspace(B).loss, epoch=eng.capture_epoch)
            if self.world > 1 or self.force:
                # first replay of a graph that holds RCCL kernels, under the watchdog of the replicated trainer (a capture problem that only
                # shows with real peers would hang inside the device queue forever); the state is restored afterwards
                DataParallelTrainer._guarded_first_replay(self, g, state)
                self.sync_parameters() if self.pending else None
                for t, sv in zip(state, snap):
                    t.copy_(sv)
                eng.refresh_weight_planes()
                self.pending = False
                torch.cuda.synchronize()
            return self._cap
        graphs = {}

        def capture_phase(i, fn):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            graphs[i] = g
            g.replay()                                                  # the capture itself runs nothing: keep the step's data flow intact
        self._step_body(sx, sy, weight, capture_phase)
        self.sync_parameters()
        torch.cuda.synchronize()
        for t, sv in zip(state, snap):
            t.copy_(sv)
        eng.refresh_weight_planes()
        self.pending = False
        torch.cuda.synchronize()
        self._cap = dict(B=B, graphs=graphs, x=sx, y=sy, weight=weight, loss=eng.workspace(B).loss, epoch=eng.capture_epoch)
        return self._cap

    def step_graph(self):
        cap = self._cap
        if cap['epoch'] != self.eng.capture_epoch:
            raise RuntimeError('the captured step is stale (set_dropout changed a value baked into the graphs): capture() again')
        if 'whole' in cap:
            cap['whole'].replay()
            self.pending = True
            return cap['loss'][0]
        self._step_body(cap['x'], cap['y'], cap.get('weight'), lambda i, fn: cap['graphs'][i].replay())
        return cap['loss'][0]

    def step(self, x, y, weight=None):
        if not self.use_graphs:
            return self.step_eager(x, y, weight)
        if self._cap is None or self._cap['B'] != x.shape[0] or self._cap['epoch'] != self.eng.capture_epoch:
            self.capture(x.shape[0], weight)
        self._cap['x'].copy_(x, non_blocking=True)
        self._cap['y'].copy_(y, non_blocking=True)
        return self.step_graph()


class PointDataParallelTrainer:
    """Data-parallel step of the point path (cfg-4 on 4 GPUs, cfg-5 on 8): replicated parameters, each rank
    trains on its own clouds, gradients are averaged once per step, SGD+momentum on every rank.  The reference's point
    trainers are single-GPU, so the contract is: per-replica arithmetic == the single-GPU path, BatchNorm statistics stay
    replica-local (no SyncBN; DDP's constructor broadcast still copies rank 0's parameters AND running statistics once).

    The backward runs in two halves (PointEngine.backward_top / backward_bottom): when the top half returns, the gradient
    slice [grad_split, numel) -- transformer blocks, TransitionUps, heads: 97 % of the parameters -- is final and its
    all-reduce is in flight on RCCL's stream while the TransitionDown half computes; the small head slice follows.  The
    1/world averaging is folded into the fused SGD kernel (grad_scale).  With HIP graphs: [forward, CE, top] | RCCL |
    [bottom] | RCCL | [SGD], i.e. three replays and two collectives per step."""

    def __init__(self, engine, group=None, use_graphs=True, force_collectives=False):
        self.eng, self.group = engine, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        broadcast_parameters(engine.arena.p, 0, group)
        for buf in engine.bn_buffers():
            broadcast_parameters(buf, 0, group)
        engine.refresh_weight_planes()
        split, n = engine.grad_split(), engine.arena.g.numel()
        self.slices = [(split, n), (0, split)] if 0 < split < n else [(0, n)]
        self.reducer = BucketedGradReducer(engine.arena.g, self.slices, group, force=force_collectives)
        engine.grad_scale = 1.0 / self.world
        self.use_graphs = use_graphs
        self._cap = None

    def _halves(self, B, x, y, starts):
        eng = self.eng

        def top():
            eng.forward(x, starts)
            eng.cross_entropy(B, y)
            eng.backward_top(B)

        return [top, lambda: eng.backward_bottom(B)]

    def step_eager(self, x, y, starts):
        eng, B = self.eng, x.shape[0]
        halves = self._halves(B, x, y, starts)
        for k, fn in enumerate(halves):
            fn()
            if k < len(self.slices):
                self.reducer.launch(k)
        self.reducer.wait()
        loss = eng.workspace(B).loss[0]
        eng.sgd_step()
        return loss

    def capture(self, x, y, starts):
        """Captures the three graphs over the given static buffers (copy new batches / FPS starts into them, then step_graph())."""
        eng, B = self.eng, x.shape[0]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with eng._preserved_state():                                    # parameters, momentum, step flag, BN statistics restored
            with torch.cuda.stream(side):                               # warm-up: kernel attributes, workspaces
                for fn in self._halves(B, x, y, starts):
                    fn()
                eng.sgd_step()
            torch.cuda.current_stream().wait_stream(side)
        graphs = []
        for fn in self._halves(B, x, y, starts):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            graphs.append(g)
        g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_opt):
            eng.sgd_step()
        self._cap = dict(B=B, graphs=graphs, opt=g_opt, x=x, y=y, starts=starts, loss=eng.workspace(B).loss)
        return self._cap

    def step_graph(self):
        cap = self._cap
        for k, g in enumerate(cap['graphs']):
            g.replay()
            if k < len(self.slices):
                self.reducer.launch(k)
        self.reducer.wait()
        cap['opt'].replay()
        return cap['loss'][0]

    # ---- geometry one step ahead (default for steady-state training) -------------------------------------------------------
    def capture_pipelined(self, x, y, starts):
        """Like capture(), but FPS / kNN / neighbour tables of the NEXT batch are computed on the engine's side stream while the
        current batch trains (PointEngine.train_step_pipelined): two buffer sets p = 0, 1, graphs [top_p | bottom_p] that train on
        set p with geometry set p and prepare geometry set 1-p from buffers 1-p.  Level 0's FPS alone is 0.5 - 1.2 ms of mostly
        idle GPU when it runs inside the step (32 - 128 workgroups).  Protocol: prime(x, y, starts) once, then
        step_pipelined(next_x, next_y, next_starts) per step -- it trains on the batch submitted one call earlier."""
        eng, B = self.eng, x.shape[0]
        xs, ys = [x.clone(), x.clone()], [y.clone(), y.clone()]
        sts = [tuple(s.clone() for s in starts), tuple(s.clone() for s in starts)]
        ws = eng.workspace(B)

        def halves(p):
            def top():
                cur = ws.geo[p]
                if ws.geo[1 - p] is None:
                    ws.geo[1 - p] = eng._new_geometry(ws, B)
                nxt = ws.geo[1 - p]
                eng._geometry(ws, B, xs[1 - p], sts[1 - p], nxt)          # side stream: forks here ...
                eng.forward(xs[p], sts[p], geometry=cur)
                eng.cross_entropy(B, ys[p])
                eng.backward_top(B)
                if nxt.events[-1] is not None:
                    torch.cuda.current_stream().wait_event(nxt.events[-1])  # ... joins inside the same graph

            def bottom():
                eng._activate(ws, ws.geo[p])
                eng.backward_bottom(B)
            return [top, bottom]

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with eng._preserved_state():
            with torch.cuda.stream(side):                               # warm-up: kernel attributes, workspaces, both geometry sets
                eng.prepare_geometry(xs[0], sts[0], 0)
                for p in (0, 1):
                    for fn in halves(p):
                        fn()
                    eng.sgd_step()
            torch.cuda.current_stream().wait_stream(side)
        graphs = []
        for p in (0, 1):
            gp = []
            for fn in halves(p):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    fn()
                gp.append(g)
            graphs.append(gp)
        g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_opt):
            eng.sgd_step()
        self._pcap = dict(B=B, graphs=graphs, opt=g_opt, xs=xs, ys=ys, sts=sts, loss=ws.loss, p=0, primed=False)
        return self._pcap

    def prime(self, x, y, starts):
        """Submits the first batch: its geometry is computed now (not overlapped with anything)."""
        if getattr(self, '_pcap', None) is None or self._pcap['B'] != x.shape[0]:
            self.capture_pipelined(x, y, starts)
        c = self._pcap
        p = c['p']
        c['xs'][p].copy_(x); c['ys'][p].copy_(y)
        for d, s in zip(c['sts'][p], starts):
            d.copy_(s)
        self.eng.prepare_geometry(c['xs'][p], c['sts'][p], p)
        c['primed'] = True

    def step_pipelined(self, next_x, next_y, next_starts):
        """One training step on the batch submitted by the previous call (or prime()); the geometry of (next_x, next_starts) is
        prepared meanwhile.  Returns the loss of the batch that was trained."""
        c = self._pcap
        assert c is not None and c['primed'], 'prime(x, y, starts) first'
        p = c['p']
        c['xs'][1 - p].copy_(next_x, non_blocking=True); c['ys'][1 - p].copy_(next_y, non_blocking=True)
        for d, s in zip(c['sts'][1 - p], next_starts):
            d.copy_(s, non_blocking=True)
        for k, g in enumerate(c['graphs'][p]):
            g.replay()
            if k < len(self.slices):
                self.reducer.launch(k)
        self.reducer.wait()
        c['opt'].replay()
        c['p'] = 1 - p
        return c['loss'][0]

    def step(self, x, y, starts):
        if not self.use_graphs:
            return self.step_eager(x, y, starts)
        if self._cap is None or self._cap['B'] != x.shape[0]:
            sx, sy, ss = x.clone(), y.clone(), tuple(s.clone() for s in starts)
            self.capture(sx, sy, ss)
        cap = self._cap
        cap['x'].copy_(x, non_blocking=True)
        cap['y'].copy_(y, non_blocking=True)
        for d, s in zip(cap['starts'], starts):
            d.copy_(s, non_blocking=True)
        return self.step_graph()
