"""The bench workload's training DYNAMICS on the CPU oracle: cfg-2 (deit_small + VoxelEmbed 32^3, batch 64), ONE repeated synthetic batch, the
reference's Adam at lr = 1e-3 (README recipe) -- what bench.py's GPU run does for --steps + --warmup steps.  Prints the loss every step so that
the bench line's loss_last_step (0.0003 .. 3.3 between runs) can be read against the reference's own behaviour (tools only).
    python tools/r6/oracle_single_batch_dynamics.py [steps=260] [threads=8]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import voxel_oracle as vo
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 260
torch.set_num_threads(int(sys.argv[2]) if len(sys.argv) > 2 else 8)
cfg = dict(backbone='deit_small_patch16_224', embed_layer='VoxelEmbed', cell=6, patch=5, n_classes=40)
sd = vo.init_state_dict(seed=9, voxel_size=32, pos_embedding='default', **cfg)
names = vo.used_param_names(sd, 'default')
m = {k: torch.zeros_like(sd[k]) for k in names}; v = {k: torch.zeros_like(sd[k]) for k in names}
x, y = vo.synthetic_batch(64, 32, 40, seed=9)
kw = {k: cfg[k] for k in ('backbone', 'embed_layer', 'cell', 'patch')}
losses, t0 = [], time.time()
for step in range(1, steps + 1):
    _, loss, grads = vo.loss_and_grads(sd, x, y, **kw)
    for k, g in grads.items(): vo.adam_step(sd[k], g, m[k], v[k], step)
    losses.append(float(loss))
    if step % 10 == 0 or losses[-1] > 10 * min(losses):
        print(f'step {step:4d}  loss {losses[-1]:.5f}   (min so far {min(losses):.5f})  [{time.time() - t0:.0f} s]', flush=True)
spikes = [i + 1 for i in range(1, len(losses)) if losses[i] > 10 * min(losses[:i]) and losses[i] > 0.1]
print(f'steps with loss > 10 x the running minimum (and > 0.1): {spikes[:40]}')
