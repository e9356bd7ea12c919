"""Per-epoch hyper-parameter schedules of the reference trainers, restated as closed forms (host helpers: the values are
written into the engines' device-resident optimizer state -- VoxelEngine.set_lr, PointEngine.set_lr / set_bn_momentum -- so
captured HIP graphs follow them without re-capture).

  voxel_lr          train_cls_voxel.py:195-198,293-294: Adam + StepLR(step_size, gamma) stepped with an EXPLICIT epoch
                    (scheduler.step(scheduler.last_epoch + 1) -> torch's closed form base * gamma ** (epoch // step_size)),
                    then pytorch_warmup.UntunedLinearWarmup.dampen() once per EPOCH (the reference calls it per epoch, not per
                    iteration): lr *= min(1, (k + 1) / warmup_period) with warmup_period = int(2 / (1 - beta2)) (= 1999 for beta2 = 0.999 in double arithmetic, as the library computes it) and
                    k = number of dampen() calls so far -- the warm-up constructor already dampens once (k = 0), so epoch e
                    trains with base * gamma ** (e // step_size) * min(1, (e + 1) / warmup_period).
  point_cls_lr      train_cls.py:93,128: StepLR(step_size=50, gamma=0.3) stepped once per epoch.
  partseg_lr        train_partseg.py:121-125: max(lr * lr_decay ** (epoch // step_size), 1e-5).
  partseg_bn_momentum  train_partseg.py:126-130: max(0.9 * 0.5 ** (epoch // step_size), 0.01) applied to every BatchNorm
                    (MOMENTUM_ORIGINAL = 0.9 is the reference's constant, :102-105)."""


def voxel_lr(epoch, base_lr=0.05, step_size=20, gamma=0.5, beta2=0.999, warmup=True):
    lr = base_lr * gamma ** (epoch // int(step_size))
    if warmup:
        period = int(2.0 / (1.0 - beta2))
        lr *= min(1.0, (epoch + 1) / period)
    return lr


def point_cls_lr(epoch, base_lr=0.01, step_size=50, gamma=0.3):
    return base_lr * gamma ** (epoch // step_size)


def partseg_lr(epoch, base_lr=0.05, lr_decay=0.5, step_size=20, clip=1e-5):
    return max(base_lr * lr_decay ** (epoch // step_size), clip)


def partseg_bn_momentum(epoch, step_size=20, original=0.9, decay=0.5, floor=0.01):
    return max(original * decay ** (epoch // step_size), floor)


class EpochSchedule:
    """Applies the schedules of one reference trainer to an engine at the top of every epoch:
        sched = EpochSchedule.for_voxel(engine, base_lr=0.05); for epoch in ...: sched.begin_epoch(epoch); <steps>"""

    def __init__(self, engine, lr_fn, bn_fn=None):
        self.engine, self.lr_fn, self.bn_fn = engine, lr_fn, bn_fn

    @classmethod
    def for_voxel(cls, engine, **kw):
        return cls(engine, lambda e: voxel_lr(e, **kw))

    @classmethod
    def for_point_cls(cls, engine, **kw):
        return cls(engine, lambda e: point_cls_lr(e, **kw))

    @classmethod
    def for_partseg(cls, engine, base_lr=0.05, lr_decay=0.5, step_size=20):
        return cls(engine, lambda e: partseg_lr(e, base_lr, lr_decay, step_size), lambda e: partseg_bn_momentum(e, step_size))

    def begin_epoch(self, epoch):
        lr = self.lr_fn(epoch)
        self.engine.set_lr(lr)
        if self.bn_fn is not None:
            self.engine.set_bn_momentum(self.bn_fn(epoch))
        return lr
