"""timm.models.layers helpers the reference imports
(models/vit_3d_2d_pretrain.py:8, models/DeIT.py:12, models/vip_3d.py:6)."""
import collections.abc
from itertools import repeat

import torch
from torch import nn


def to_2tuple(x):
    if isinstance(x, collections.abc.Iterable) and not isinstance(x, str):
        return tuple(x)
    return tuple(repeat(x, 2))


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    # timm 0.3.2 truncates at absolute bounds [a, b] (not in units of std);
    # torch.nn.init.trunc_normal_ has the same convention.
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


class DropPath(nn.Module):
    """Stochastic depth per sample.  The reference always builds drop_path_rate=0,
    in which case timm's Block uses nn.Identity instead of this class."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if not self.training or not self.drop_prob:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x / keep * mask
