import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import simple3d_former_amd as s3d
from simple3d_former_amd import _lib as L
from oracle import bf16_backward as bb, voxel_oracle as vo
from _util import load_case, rebuild_inputs, MODEL_KEYS, fwd_kwargs
for name in sys.argv[1:] or ['cfg2_small_v32_b4', 'tiny_v12_default_b3']:
    z, cfg = load_case(name)
    sd, x, y = rebuild_inputs(cfg, z)
    _, _, ref = bb.loss_and_grads(sd, x, y, round=True, **fwd_kwargs(cfg))
    _, _, ex = vo.loss_and_grads(sd, x, y, **fwd_kwargs(cfg))
    L.lib().s3d_set_deterministic(1)
    eng = s3d.VoxelEngine(device='cuda', **{k: cfg[k] for k in MODEL_KEYS}); eng.load_state_dict(sd)
    eng.forward(x.cuda()); eng.cross_entropy(cfg['batch'], y.cuda()); eng.zero_grad(); eng.backward(cfg['batch'])
    torch.cuda.synchronize()
    rows = []
    for k, g in ref.items():
        got = eng.arena.grad(k).double().cpu().reshape(g.shape)
        rms = float(g.pow(2).mean().sqrt()) + 1e-30
        rows.append((float((got - g).pow(2).mean().sqrt()) / rms, float((got - g).abs().max()) / rms,
                     float((got - ex[k].double()).pow(2).mean().sqrt()) / rms, k))
    rows.sort(reverse=True)
    print(name, 'rms-err-vs-emulated  max-err  rms-err-vs-fp32')
    for r in rows[:14]:
        print('   %.2e  %.2e  %.2e  %s' % r)
    print('   median vs emulated %.2e, vs fp32 %.2e' % (sorted(r[0] for r in rows)[len(rows) // 2], sorted(r[2] for r in rows)[len(rows) // 2]))
