"""What does a second live branch cost a captured HIP graph on this runtime?  A chain of N small kernels (x.add_(1) on 256 KB) is
captured (a) alone, (b) with ONE extra kernel on a second stream forked before node `fork` and joined after node `join`.  Replay
time per chain node, and the difference per node that lies inside the two-branch region.  (round 3: overlapped Adam slices and
tail launches on a second stream made the cfg-2 step slower by ~2.8 us per graph node in the parallel region.)"""
import torch

N = 150
dev = 'cuda'
x = torch.zeros(65536, device=dev)
y = torch.zeros(65536, device=dev)
side = torch.cuda.Stream()


def capture(fork, join):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        for i in range(N):
            if i == fork:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    y.add_(1.0)
            x.add_(1.0)
            if i == join:
                main.wait_stream(side)
    return g


def timeit(g, reps=200):
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


x.add_(1.0); y.add_(1.0)
torch.cuda.synchronize()
base = timeit(capture(-1, -1))
print(f'chain of {N}: {base:.1f} us = {base / N:.2f} us per node')
for fork, join in ((0, N - 1), (0, 0), (0, 10), (0, 74), (75, N - 1), (140, N - 1), (N - 1, N - 1)):
    t = timeit(capture(fork, join))
    span = join - fork + 1
    print(f'branch forked before node {fork:3d}, joined after node {join:3d}: {t:.1f} us  (+{t - base:.1f} us, {(t - base) / span:+.2f} us per node in the region)')
