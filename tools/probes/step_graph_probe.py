"""Does an external event-record node in the middle of ONE captured HIP graph (a marker kernel node replaced by an event-record node: s3d_graph_events_at_markers) let
host-launched work on another stream start when its segment is done, while the rest of the graph still runs?

    python tools/probes/step_graph_probe.py

Segment 0 adds 1 to a buffer, segment 1 is a long tail of no-op passes over it.  A side stream waits for "segment 0 done", copies one
element and records an event: it must see the value of THIS replay and finish long before the whole graph does."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simple3d_former_amd import _lib as L  # noqa: E402

lib = L.lib()
a = torch.zeros(1 << 26, device='cuda'); b = torch.zeros(1, device='cuda')
side = torch.cuda.Stream()
ev = ctypes.c_void_p()
L.check(lib.s3d_event_create(ctypes.byref(ev)), 'event_create')
g = torch.cuda.CUDAGraph(keep_graph=True)
with torch.cuda.graph(g):
    a.add_(1.0)                                                                         # "segment 0"
    L.check(lib.s3d_graph_marker(0, L.current_stream()), 'marker')                     # becomes the event-record node
    for _ in range(40):                                                                 # "segment 1": a long tail
        a.mul_(1.0)
L.check(lib.s3d_graph_events_at_markers(ctypes.c_void_p(g.raw_cuda_graph()), (ctypes.c_void_p * 1)(ev), 1), 'events_at_markers')
for it in range(4):
    e0, e_side, e_all = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    L.check(lib.s3d_stream_wait_event(ctypes.c_void_p(side.cuda_stream), ev), 'wait')
    with torch.cuda.stream(side):
        b.copy_(a[:1])
        e_side.record()
    e_all.record()
    torch.cuda.synchronize()
    print(f'replay {it}: side stream saw {float(b[0]):.0f} (expected {it + 1}); side done after {e0.elapsed_time(e_side):.3f} ms, '
          f'whole graph after {e0.elapsed_time(e_all):.3f} ms')
L.check(lib.s3d_event_destroy(ev), 'destroy')
