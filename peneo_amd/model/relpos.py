"""Host-side constants of the relative-position bias (K4).

``relative_position_bucket`` of the reference (modeling_layoutlmv3.py:586-613) maps a signed
integer distance to a bucket with an fp32 ``log`` followed by truncation.  The distance is a
small integer, so the device kernel looks the unsigned part up in a table; the table is filled
here once with the *same* fp32 torch-CPU expression the reference evaluates, which makes the
bucket indices bit-identical to the reference's CPU path (a device ``logf`` may differ by an ulp
at a bucket edge).  This is a constant table, not a compute fallback.
"""
from __future__ import annotations

import math
from functools import lru_cache

import torch


@lru_cache(maxsize=None)
def _bucket_lut_cpu(num_buckets: int, max_distance: int, length: int) -> torch.Tensor:
    half = num_buckets // 2
    max_exact = half // 2
    n = torch.arange(length, dtype=torch.long)
    is_small = n < max_exact
    val_if_large = max_exact + (
        torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (half - max_exact)
    ).to(torch.long)
    val_if_large = torch.min(val_if_large, torch.full_like(val_if_large, half - 1))
    return torch.where(is_small, n, val_if_large).to(torch.uint8)


def bucket_lut(num_buckets: int, max_distance: int, length: int = 1024) -> torch.Tensor:
    """uint8 [length]: unsigned bucket of |delta| (the sign contributes ``num_buckets // 2``).
    ``length`` must exceed the largest |delta| that is not yet saturated; 1024 covers bbox
    coordinates (0..1000); positions are clamped by the kernel (saturated beyond max_distance)."""
    assert length > max_distance
    return _bucket_lut_cpu(num_buckets, max_distance, length).clone()


def visual_xy(grid: int = 14, max_len: int = 1000):
    """(x, y) of the visual tokens used by the 2-D bias: the cls box [1,1,999,999] then the patch
    grid (modeling_layoutlmv3.py:879-901); the bias uses bbox[..., 0] and bbox[..., 3] (:647-648)."""
    xs = [1]
    ys = [max_len - 1]
    for r in range(grid):
        for c in range(grid):
            xs.append(max_len * c // grid)
            ys.append(max_len * (r + 1) // grid)
    return torch.tensor(xs, dtype=torch.int32), torch.tensor(ys, dtype=torch.int32)
