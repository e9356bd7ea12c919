"""Where does the gradient error of the point path come from?  One training-mode step of cfg-4 / cfg-5 shapes at several batch sizes,
default and deterministic dispatch, every gradient against the CPU oracle: rms error / tensor rms, the regression coefficient
alpha = <got, ref> / <ref, ref> (a bias -- missing rows, a wrong scale -- moves alpha; zero-mean rounding noise does not) and, for the
first convolution of TransitionDown 0, the same per column group (xyz columns | feature columns).
    python tools/probes/point_grad_noise_probe.py cls 1024 6 40 16 128"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import point_oracle as po
from simple3d_former_amd.point_engine import PointEngine
from simple3d_former_amd import _lib as L

task, n, d, c = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
for B in [int(v) for v in sys.argv[5:]]:
    sd = po.init_state_dict(backbone='deit_tiny_patch16_224', n_classes=c, d_points=d, seed=9)
    x, y, starts = po.synthetic_points(B, n, d, c, task, seed=9)
    _, _, ref, _ = po.loss_and_grads(sd, x, y, task=task, backbone='deit_tiny_patch16_224', starts=starts)
    for det in (0, 1):
        L.lib().s3d_set_deterministic(det)
        eng = PointEngine(backbone='deit_tiny_patch16_224', n_points=n, d_points=d, n_classes=c, task=task, device='cuda')
        eng.load_state_dict(sd)
        eng.forward(x.cuda(), tuple(s.cuda() for s in starts)); eng.cross_entropy(B, y.cuda()); eng.zero_grad(); eng.backward(B)
        rows = []
        for k, r in ref.items():
            g = eng.arena.grad(k).double().cpu().reshape(-1); r = r.double().reshape(-1)
            rms = float(r.pow(2).mean().sqrt()) + 1e-30
            rows.append((float((g - r).pow(2).mean().sqrt()) / rms, k, float((g * r).sum() / (r * r).sum()), rms, r.numel()))
        rows.sort(reverse=True)
        print(f'--- {task} B={B} deterministic={det}: worst 12 of {len(rows)} tensors (rms err / rms, alpha, rms, entries)')
        for e, k, al, rms, nn in rows[:12]:
            print(f'   {e:8.4f}  alpha {al:8.5f}  rms {rms:9.3e}  n {nn:7d}  {k}')
        k = 'transition_downs.0.sa.mlp_convs.0.weight'
        g = eng.arena.grad(k).double().cpu().reshape(ref[k].shape[0], -1); r = ref[k].double().reshape(ref[k].shape[0], -1)
        for nm, sl in (('xyz cols', slice(0, 3)), ('feat cols', slice(3, None))):
            rr, ee = r[:, sl], (g - r)[:, sl]
            print(f'   td0.conv0 {nm}: ref rms {float(rr.pow(2).mean().sqrt()):.3e} err rms {float(ee.pow(2).mean().sqrt()):.3e} worst {float(ee.abs().max()):.3e}')
        if det == 1:            # per feature column of td0.conv0: does the error follow the column mean of the features (sum_rows dPf = 0 cancels it exactly only in exact arithmetic)?
            ws = eng.workspace(B)
            f = ws.f.double().cpu()                                  # [B*N, C0] input features of TransitionDown 0
            mu, sd_ = f.mean(0), f.std(0)
            e = (g - r)[:, 3:]
            print('   td0.conv0 feature columns: |mean f| / std f, ref col rms, err col rms')
            for c in range(e.shape[1]):
                print(f'      col {c:2d}: {float(mu[c].abs() / sd_[c]):6.2f}  {float(r[:, 3 + c].pow(2).mean().sqrt()):.3e}  {float(e[:, c].pow(2).mean().sqrt()):.3e}')
    L.lib().s3d_set_deterministic(0)
