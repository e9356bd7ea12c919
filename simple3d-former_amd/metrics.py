"""Evaluation bookkeeping of the reference's test loops, accumulated on the device (no per-batch logits D2H copy).

ClsEvaluator     train_cls_voxel.py:300-329 (accuracy, mean class accuracy = class_correct / class_total summed / N_CLASSES)
                 and train_cls.py:22-41 (instance accuracy; mean over the classes that occur)
PartSegEvaluator train_partseg.py:157-217 (accuracy, class_avg_accuracy, class_avg_iou, inctance_avg_iou)

The counters live in one int64 device tensor; `update()` is one kernel launch, `result()` is the only host sync."""
import ctypes

import numpy as np
import torch

from . import _lib as L

# ShapeNetPart: category -> (first part id, number of parts).  Part ids of a category are contiguous
# (train_partseg.py:26-29 holds the same table as explicit lists).
SHAPENET_PARTS = (('Airplane', 0, 4), ('Bag', 4, 2), ('Cap', 6, 2), ('Car', 8, 4), ('Chair', 12, 4), ('Earphone', 16, 3),
                  ('Guitar', 19, 3), ('Knife', 22, 2), ('Lamp', 24, 4), ('Laptop', 28, 2), ('Motorbike', 30, 6),
                  ('Mug', 36, 2), ('Pistol', 38, 3), ('Rocket', 41, 3), ('Skateboard', 44, 3), ('Table', 47, 3))


def seg_classes_dict(table=SHAPENET_PARTS):
    return {name: list(range(first, first + cnt)) for name, first, cnt in table}


class ClsEvaluator:
    def __init__(self, n_classes, device='cuda'):
        self.C = n_classes
        self.counts = torch.zeros(1 + 2 * n_classes, dtype=torch.int64, device=device)
        self.lib = L.lib()

    def reset(self):
        self.counts.zero_()

    def update(self, logits, target, ld=None, return_pred=False):
        """logits fp32 [rows, >=C] (row stride `ld`, default logits.stride(0)); target int64 [rows]."""
        assert logits.dtype == torch.float32 and logits.is_cuda and logits.stride(-1) == 1
        target = target.to(device=logits.device, dtype=torch.int64).contiguous()
        rows = target.numel()
        pred = torch.empty(rows, dtype=torch.int32, device=logits.device) if return_pred else None
        L.check(self.lib.s3d_cls_eval(L.ptr(logits), int(ld or logits.stride(0)), L.ptr(target), ctypes.c_long(rows), self.C,
                                      L.ptr(pred), L.ptr(self.counts), L.current_stream()), 'cls_eval')
        return pred

    def result(self):
        c = self.counts.cpu().numpy()
        correct, cls_correct, cls_total = int(c[0]), c[1:1 + self.C], c[1 + self.C:]
        total = int(cls_total.sum())
        with np.errstate(divide='ignore', invalid='ignore'):
            acc = cls_correct / cls_total.astype(np.float64)
        seen = cls_total > 0
        return {'accuracy': correct / float(total), 'mean_class_accuracy': float(acc.sum() / self.C),
                'mean_seen_class_accuracy': float(acc[seen].mean()) if seen.any() else float('nan'),
                'total': total, 'class_correct': cls_correct.copy(), 'class_total': cls_total.copy()}


class PartSegEvaluator:
    def __init__(self, num_part=50, table=SHAPENET_PARTS, device='cuda'):
        self.P = num_part
        self.table = tuple(table)
        rng = np.zeros((num_part, 2), dtype=np.int32)
        for _, first, cnt in table:
            assert cnt <= 16, 'at most 16 parts per category'
            rng[first:first + cnt] = (first, cnt)
        assert (rng[:, 1] > 0).all(), 'every part label must belong to a category'
        self.part_range = torch.from_numpy(rng).to(device)
        self.counts = torch.zeros(1 + 2 * num_part, dtype=torch.int64, device=device)
        self._iou, self._first = [], []
        self.lib = L.lib()

    def reset(self):
        self.counts.zero_()
        self._iou, self._first = [], []

    def update(self, logits, target, ld=None, return_pred=False):
        """logits fp32 [B, N, >=P] (row stride `ld`); target int64 [B, N]."""
        assert logits.dtype == torch.float32 and logits.is_cuda and logits.stride(-1) == 1
        B, N = target.shape
        target = target.to(device=logits.device, dtype=torch.int64).contiguous()
        iou = torch.empty(B, dtype=torch.float64, device=logits.device)
        first = torch.empty(B, dtype=torch.int32, device=logits.device)
        pred = torch.empty(B, N, dtype=torch.int32, device=logits.device) if return_pred else None
        L.check(self.lib.s3d_partseg_eval(L.ptr(logits), int(ld or logits.stride(-2)), L.ptr(target), B, N, self.P,
                                          L.ptr(self.part_range), L.ptr(pred), L.ptr(iou), L.ptr(first), L.ptr(self.counts),
                                          L.current_stream()), 'partseg_eval')
        self._iou.append(iou)
        self._first.append(first)
        return pred

    def result(self):
        c = self.counts.cpu().numpy()
        correct, cls_correct, cls_seen = int(c[0]), c[1:1 + self.P], c[1 + self.P:]
        iou = torch.cat(self._iou).cpu().numpy()
        first = torch.cat(self._first).cpu().numpy()
        with np.errstate(divide='ignore', invalid='ignore'):
            per_cat = {name: float(np.mean(iou[first == f])) if (first == f).any() else float('nan') for name, f, _ in self.table}
            class_avg_acc = float(np.mean(cls_correct / cls_seen.astype(np.float64)))
        return {'accuracy': correct / float(int(cls_seen.sum())), 'class_avg_accuracy': class_avg_acc,
                'class_avg_iou': float(np.mean(list(per_cat.values()))), 'inctance_avg_iou': float(np.mean(iou)),
                'shape_ious': iou, 'category_iou': per_cat}
