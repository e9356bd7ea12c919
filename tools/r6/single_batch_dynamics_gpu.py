"""The bench workload's training dynamics on the GPU (cfg-2, ONE repeated batch, Adam lr 1e-3 -- what bench.py runs for warmup + steps steps),
several independent runs per backward precision: the step at which the loss leaves the 3.38 plateau (= entropy of the batch's label histogram) and
the loss at steps 240 / 300, next to the CPU oracle's trajectory (tools/r6/oracle_single_batch_dynamics.py: plateau until ~150, < 1 at step 171,
1e-3 at 250).  Explains bench.py's loss_last_step (tools only).     python tools/r6/single_batch_dynamics_gpu.py [runs=6] [steps=300]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import simple3d_former_amd as s3d
from oracle import voxel_oracle as vo
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
CFG = dict(backbone='deit_small_patch16_224', embed_layer='VoxelEmbed', voxel_size=32, cell=6, patch=5, n_classes=40)
sd = vo.init_state_dict(seed=9, pos_embedding='default', **CFG)
x, y = vo.synthetic_batch(64, 32, 40, seed=9)
x, y = x.cuda(), y.cuda()
for mode in ('bf16', 'split'):
    for r in range(runs):
        eng = s3d.VoxelEngine(device='cuda', pos_embedding='default', backward=mode, **CFG)
        eng.load_state_dict(sd); eng.set_optimizer(lr=1e-3)
        losses = [float(eng.train_step(x, y)) for _ in range(steps)]
        esc = next((i + 1 for i, l in enumerate(losses) if l < 1.0), None)
        back = [i + 1 for i in range(1, steps) if esc and i + 1 > esc and losses[i] > 1.0]
        print(f'backward {mode:5s} run {r}: loss at 60 {losses[59]:.5f}  leaves the plateau (< 1) at step {esc}  loss at 240 {losses[239]:.5f}  at {steps} {losses[-1]:.5f}'
              + (f'  back above 1 at steps {back[:6]}' if back else ''), flush=True)
        del eng
