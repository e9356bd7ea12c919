import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import tests.test_gpu_dp_two_ranks as T
import torch.multiprocessing as mp
if __name__ == '__main__':
    world, port = 2, T._free_port()
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    procs = [ctx.Process(target=T._worker_wire, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    [p.join() for p in procs]
    for rank in range(2):
        lf, lb = res[rank][1]["fp32"][0], res[rank][1]["bf16"][0]
        print("rank", rank, "fp32", [round(v, 4) for v in lf])
        print("rank", rank, "rel dev", ["%.1e" % (abs(a - b) / abs(b)) for a, b in zip(lb, lf)])
