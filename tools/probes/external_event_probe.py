"""Does this stack (torch 2.10 + ROCm) support EXTERNAL events inside a captured HIP graph (hipEventRecordExternal)?  If so, one
graph can hold the whole backward while host-launched collectives on another stream wait for events recorded in its middle."""
import torch

try:
    ev = torch.cuda.Event(external=True)
except TypeError as e:
    print('no external= keyword:', e); raise SystemExit
a = torch.zeros(1 << 24, device='cuda'); b = torch.zeros(1, device='cuda')
side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        a.add_(1.0)                       # "segment 0"
        ev.record()                       # external event in the middle of the graph
        for _ in range(50):
            a.mul_(1.0)                   # "segment 1": long tail after the event
    for it in range(3):
        g.replay()
        with torch.cuda.stream(side):
            side.wait_event(ev)           # host-launched work that depends on the mid-graph event only
            b.copy_(a[:1])                # must see a == it + 1 (segment 0 of THIS replay done), without waiting for the tail
        torch.cuda.synchronize()
        print('replay', it, 'side stream saw', float(b[0]), 'expected', float(it + 1))
except Exception as e:                    # noqa: BLE001
    print('failed:', type(e).__name__, str(e)[:300])
