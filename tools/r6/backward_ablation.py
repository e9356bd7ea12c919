"""GPU ablation: the stable-regime trained fixture (tests/golden/trained_stable_cfg1_small_v30_adam400.npz, five
reference seeds) replayed by the fused HIP training step in every backward-precision mode; prints one table.
    python tools/r6/backward_ablation.py [modes, comma separated] [runs per seed]"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import simple3d_former_amd as s3d
from oracle import voxel_oracle as vo

DEV = 'cuda'


def run(z, cfg, seed, mode, det=False):
    kw = {k: cfg[k] for k in ('backbone', 'embed_layer', 'voxel_size', 'cell', 'patch', 'n_classes', 'pos_embedding', 'head')}
    sd = vo.init_state_dict(seed=seed, exercise_all=False, portable=True, **kw)
    ekw = dict(precise_backward=True) if mode == 'precise' else ({} if mode == 'bf16' else dict(backward=mode))
    eng = s3d.VoxelEngine(device=DEV, lr=cfg['lr'], **ekw, **kw)
    eng.load_state_dict(sd)
    dk = dict(base=cfg['density_base'], step=cfg['density_step'])
    data = [vo.synthetic_class_batch(cfg['batch'], cfg['voxel_size'], cfg['cell'], cfg['labels'], seed=500 + i, **dk) for i in range(cfg['n_batches'])]
    data = [(x.to(DEV), y.to(DEV)) for x, y in data]
    xh, yh = vo.synthetic_class_batch(cfg['held_batch'], cfg['voxel_size'], cfg['cell'], cfg['labels'], seed=999, **dk)
    xh = xh.to(DEV)
    losses, accs = [], []
    for step in range(cfg['steps']):
        x, y = data[step % len(data)]
        losses.append(eng.train_step(x, y).clone())
        if step + 1 in cfg['checkpoints']:
            am = eng.forward(xh).argmax(1).cpu()
            accs.append(float((am == yh).float().mean()))
    losses = torch.stack(losses).cpu().numpy()
    return losses, np.array(accs)


def main():
    modes = (sys.argv[1] if len(sys.argv) > 1 else 'bf16,precise').split(',')
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'trained_stable_cfg1_small_v30_adam400.npz'))
    cfg = json.loads(str(z['cfg']))
    if os.environ.get('FIXTURE') == 'adam60':      # round 5's unstable-regime fixture (Adam lr 1e-3 from the initialisation, 60 steps), more seeds
        cfg.update(steps=60, lr=1e-3, n_batches=4, labels=cfg['labels'][:9], density_base=0.04, density_step=0.07, held_batch=256, checkpoints=[40, 50, 60], tail=10)
    seeds = cfg['seeds']
    if os.environ.get('SEEDS'):
        a, b = os.environ['SEEDS'].split('-')
        seeds = list(range(int(a), int(b) + 1))
    T = cfg['tail']
    print('reference (torch.optim.Adam on the reference model, CPU fp32):')
    ref_acc, ref_tail = [], []
    for s in cfg['seeds']:
        if f'losses_{s}' not in z.files:
            continue
        l, a = z[f'losses_{s}'], z[f'held_acc_{s}']
        if os.environ.get('FIXTURE') == 'adam60':
            break
        ref_acc.append(a.mean()); ref_tail.append(np.median(l[-T:]))
        print(f'  seed {s}: held-out accuracy {a.mean():.3f}  (checkpoints {" ".join(f"{v:.3f}" for v in a)})  tail loss median {np.median(l[-T:]):.4f} mean {l[-T:].mean():.4f}')
    if ref_acc: print(f'  spread: accuracy {min(ref_acc):.3f} .. {max(ref_acc):.3f} (mean {np.mean(ref_acc):.3f}, std {np.std(ref_acc, ddof=1):.3f}); tail median {min(ref_tail):.4f} .. {max(ref_tail):.4f}')
    for mode in modes:
        accs, tails = [], []
        t0 = time.time()
        for s in seeds:
            for r in range(runs):
                l, a = run(z, cfg, s, mode)
                ref = z[f'losses_{s}'][:cfg['steps']] if f'losses_{s}' in z.files else l
                rel = np.abs(l - ref) / np.maximum(np.abs(ref), 1e-6)
                accs.append(a.mean()); tails.append(np.median(l[-T:]))
                print(f'  {mode:12s} seed {s} run {r}: held-out accuracy {a.mean():.3f} ({" ".join(f"{v:.3f}" for v in a)}) tail loss median {np.median(l[-T:]):.4f} mean {l[-T:].mean():.4f}; '
                      f'loss vs the reference of that seed: steps 0-19 {rel[:20].max():.1e}, 0-99 {rel[:100].max():.1e}', flush=True)
        print(f'{mode:12s}: accuracy {min(accs):.3f} .. {max(accs):.3f} (mean {np.mean(accs):.4f} +- {np.std(accs, ddof=1) / np.sqrt(len(accs)):.4f}, std {np.std(accs, ddof=1):.3f}, n {len(accs)}); tail median {min(tails):.4f} .. {max(tails):.4f} (geo-mean {np.exp(np.mean(np.log(tails))):.4f})  [{time.time() - t0:.0f} s]', flush=True)


if __name__ == '__main__':
    main()
