"""Margins of test_benched_backward_reproduces_the_reference_trained_accuracy_and_final_loss: the asserted statistics over repeated runs."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tests import test_gpu_trajectory as T
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    cfg, runs = T._stable_fixture_runs('bf16')
    acc = np.array([r['acc'] for r in runs]); ref = np.array([r['ref_acc'] for r in runs])
    lt = np.log([r['tail'] for r in runs]); rlt = np.log([r['ref_tail'] for r in runs])
    n = len(runs)
    se = np.sqrt(ref.var(ddof=1) / n + acc.var(ddof=1) / n); sel = np.sqrt(rlt.var(ddof=1) / n + lt.var(ddof=1) / n)
    print(f'run {rep}: steps 0-19 {max(r["rel"][:20].max() for r in runs):.1e}  0-49 {max(r["rel"][:50].max() for r in runs):.1e}  0-99 {max(r["rel"][:100].max() for r in runs):.1e}; '
          f'acc {acc.min():.3f}..{acc.max():.3f} mean {acc.mean():.4f} (ref {ref.mean():.4f}; |diff| / se = {abs(acc.mean() - ref.mean()) / se:.2f}); '
          f'log-loss diff / se = {abs(lt.mean() - rlt.mean()) / sel:.2f}', flush=True)
