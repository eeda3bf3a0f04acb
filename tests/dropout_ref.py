"""Host restatement (numpy, 64-bit arithmetic masked to 32 bits) of the counter-based mask of the classifier dropout
(peneo_amd/csrc/common.h: pair_drop_key / pair_drop_seed / pair_drop_step): keep(b, p, n) for document b, packed pair index p, hidden
column n of the [B, P, nh * D] classifier hidden (model/peneo_decoder.py:261).  Test infrastructure."""
import numpy as np
import torch

_M = np.uint64(0xFFFFFFFF)
_u = np.uint64


def _mix32(x):
    x = x.astype(np.uint64)
    x ^= x >> _u(16); x = (x * _u(0x7FEB352D)) & _M; x ^= x >> _u(15); x = (x * _u(0x846CA68B)) & _M; x ^= x >> _u(16)
    return x


def _mul24(a, k):
    return ((a & _u(0xFFFFFF)) * _u(k)) & _M


def k12_threshold(p: float) -> int:
    return int(p * 65536.0 + 0.5) if p > 0 else 0


def k12_scale(p: float) -> float:
    t = k12_threshold(p)
    return 65536.0 / (65536.0 - t) if t else 1.0


def k12_keep(seed: int, b: int, p0: int, p1: int, ncol: int, p: float) -> torch.Tensor:
    """bool [p1 - p0, ncol]: True where the hidden unit is kept."""
    thr = k12_threshold(p)
    if thr == 0:
        return torch.ones((p1 - p0, ncol), dtype=torch.bool)
    nslab = ncol // 32
    key = _mix32(np.array([(seed ^ (((b + 1) * 0x9E3779B9) & 0xFFFFFFFF)) & 0xFFFFFFFF], dtype=np.uint64))[0]
    pp = np.arange(p0, p1, dtype=np.uint64)[:, None, None]
    sl = np.arange(nslab, dtype=np.uint64)[None, :, None]
    hh = np.arange(2, dtype=np.uint64)[None, None, :]
    x = ((((pp * _u(nslab) + sl) * _u(2) + hh)) & _M) ^ key
    x = (x * _u(0x9E3779B1)) & _M     # full 32-bit multiply in front of the 24-bit rounds (common.h: pair_drop_premix)
    x ^= x >> _u(16)
    inc = _u(0x9E3779) ^ ((x >> _u(24)) << _u(8))     # the chain's increment: the 8 premixed bits the 24-bit mixer does not read
    x = _mul24(x, 0x9E3779); x ^= x >> _u(13); x = _mul24(x, 0x85EBCB); x ^= x >> _u(16)
    out = np.zeros((p1 - p0, nslab, 32), dtype=bool)
    st = x
    for i in range(16):
        st = (_mul24(st, 0xC2B2AF) + inc) & _M
        g, e = i >> 2, i & 3
        for h in range(2):
            out[:, :, 8 * g + 4 * h + e] = (st[:, :, h] >> _u(16)) >= thr
    return torch.from_numpy(out.reshape(p1 - p0, ncol))
