"""DropPath / to_2tuple / trunc_normal_ as documented for timm==0.9.12 (see package docstring)."""
import torch
import torch.nn as nn


class DropPath(nn.Module):
    """Stochastic depth: per-sample Bernoulli(keep) mask scaled by 1/keep; identity in eval."""

    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob = float(drop_prob)
        self.scale_by_keep = scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask


def to_2tuple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)
