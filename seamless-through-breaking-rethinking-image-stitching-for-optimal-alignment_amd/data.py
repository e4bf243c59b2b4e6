"""Deterministic, integer-valued synthetic image pairs (the workload generator of bench.py, smoke() and the tests).

Everything is integer arithmetic on numpy's MT19937 stream, so the same bytes come out on every
host; pairs are float32 [B,3,H,W] in 0..255 like the reference's loaders (out.py:129-146)."""
from __future__ import annotations

import numpy as np
import torch


def _box(a, r):
    """integer box blur, radius r, edge-replicated, on the last two axes."""
    for ax in (-2, -1):
        p = np.concatenate([np.repeat(a.take([0], ax), r, ax), a, np.repeat(a.take([-1], ax), r, ax)], ax)
        c = np.cumsum(p.astype(np.int64), ax)
        z = np.zeros_like(c.take([0], ax))
        c = np.concatenate([z, c], ax)
        n = a.shape[ax]
        hi = c.take(np.arange(2 * r + 1, 2 * r + 1 + n), ax)
        lo = c.take(np.arange(0, n), ax)
        a = (hi - lo) // (2 * r + 1)
    return a


def structured_pair(h=512, w=512, seed=7, shift=(5, -9), batch=1):
    """img1 = blurred blocky texture; img2 = img1 rolled by ``shift`` + small integer noise."""
    rs = np.random.RandomState(seed)
    out1, out2 = [], []
    for _ in range(batch):
        cell = 16
        base = rs.randint(0, 256, size=(3, (h + cell - 1) // cell, (w + cell - 1) // cell))
        img = np.repeat(np.repeat(base, cell, 1), cell, 2)[:, :h, :w]
        img = _box(_box(img, 4), 4)
        fine = rs.randint(-12, 13, size=(3, h, w))
        img1 = np.clip(img + _box(fine, 1), 0, 255)
        img2 = np.clip(np.roll(img1, shift, (1, 2)) + rs.randint(-3, 4, size=(3, h, w)), 0, 255)
        out1.append(img1)
        out2.append(img2)
    a = torch.from_numpy(np.stack(out1).astype(np.float32))
    b = torch.from_numpy(np.stack(out2).astype(np.float32))
    return a, b


def noise_pair(h=512, w=512, seed=1234, batch=1):
    """uniform integer noise pair (stand-in for out.py:7-8's torch.rand*255 inputs)."""
    rs = np.random.RandomState(seed)
    a = rs.randint(0, 256, size=(batch, 3, h, w)).astype(np.float32)
    b = rs.randint(0, 256, size=(batch, 3, h, w)).astype(np.float32)
    return torch.from_numpy(a), torch.from_numpy(b)
