#!/usr/bin/env python3
"""Statistics of the attention-dropout mask function (attention.hip: attn_mix24 / attn_pairhash) against mix32, in numpy.

One hash serves the two keys (2j, 2j+1) of a query row (low / high 16 bits); the mixer uses 24-bit multiplies.  Prints keep
rate (even / odd keys), bit balance of the hash, correlations of the keep decisions between the two halves of one hash,
neighbouring hashes, neighbouring rows and a few lags, and the spread of per-row / per-key keep counts against the binomial
expectation.  CPU only."""
import numpy as np

M = np.uint64(0xFFFFFFFF)
u = np.uint64


def mix32(x):
    x = x.astype(np.uint64)
    x ^= x >> u(16); x = (x * u(0x7FEB352D)) & M; x ^= x >> u(15); x = (x * u(0x846CA68B)) & M; x ^= x >> u(16)
    return x


def mix24(x):
    x = x.astype(np.uint64)
    x ^= x >> u(16); x = ((x & u(0xFFFFFF)) * u(0x9E3779)) & M
    x ^= x >> u(13); x = ((x & u(0xFFFFFF)) * u(0x85EBCB)) & M
    x ^= x >> u(16)
    return x


def main(p=0.1, T=709, rows=2000, seed=12345):
    rid = np.arange(rows, dtype=np.uint64)
    rh = mix32(u(seed) ^ ((rid * u(0x9E3779B9)) & M))
    keys = np.arange(0, T + 1, dtype=np.uint64)
    th = round(p * 65536)
    for name, fn in (("mix24", mix24), ("mix32", mix32)):
        h = fn(rh[:, None] ^ (keys >> u(1))[None, :])
        v = np.where((keys & u(1))[None, :] == 1, h >> u(16), h & u(0xFFFF))
        keep = v >= th
        k = keep.astype(float) - keep.mean()
        c = lambda a, b: float((a * b).mean() / k.var())
        bits = [float(((h >> u(b)) & u(1)).mean()) for b in range(32)]
        print(f"{name}: keep {keep.mean():.5f} (even {keep[:, 0::2].mean():.5f}, odd {keep[:, 1::2].mean():.5f}), want {1 - th / 65536:.5f}")
        print(f"   hash bit means {min(bits):.4f} .. {max(bits):.4f}")
        print(f"   corr: halves of one hash {c(k[:, 0:-1:2], k[:, 1::2]):+.4f}, neighbouring hashes {c(k[:, 1:-1:2], k[:, 2::2]):+.4f}, "
              f"rows {c(k[:-1], k[1:]):+.4f}, lags 2/4/8/32 " + " ".join(f"{c(k[:, :-l], k[:, l:]):+.4f}" for l in (2, 4, 8, 32)))
        print(f"   keep-count std per row {keep.sum(1).std():.2f} (binomial {np.sqrt((T + 1) * p * (1 - p)):.2f}), "
              f"per key {keep.sum(0).std():.2f} (binomial {np.sqrt(rows * p * (1 - p)):.2f});  noise floor of a correlation ~ {1 / np.sqrt(k.size):.4f}")


def k12(p=0.1, P=3000, nslab=60, seed=12345):
    """The classifier-dropout mask (common.h: pair_drop_seed / pair_drop_step): one mix24 seed per (pair, slab, half), 16
    fields by a 24-bit multiply-add chain.  Keep rate, correlations along the chain (lag 1..3), across halves (lag 4),
    groups (lag 8), slabs (lag 32) and pairs, joint probabilities of neighbours, count statistics."""
    key = int(mix32(np.array([(seed ^ 0x9E3779B9) & 0xFFFFFFFF], dtype=np.uint64))[0])
    pp = np.arange(P, dtype=np.uint64)[:, None, None]; sl = np.arange(nslab, dtype=np.uint64)[None, :, None]
    hh = np.arange(2, dtype=np.uint64)[None, None, :]
    st = mix24((((pp * u(nslab) + sl) * u(2) + hh) & M) ^ u(key))
    f = np.zeros((P, nslab, 32), dtype=np.uint64)
    for i in range(16):
        st = (((st & u(0xFFFFFF)) * u(0xC2B2AF)) + u(0x9E3779)) & M
        for h in range(2):
            f[:, :, 8 * (i >> 2) + 4 * h + (i & 3)] = st[:, :, h] >> u(16)
    f = f.reshape(P, nslab * 32)
    th = round(p * 65536)
    keep = f >= th
    k = keep.astype(float) - keep.mean()
    c = lambda a, b: float((a * b).mean() / k.var())
    print(f"K12 chain: keep {keep.mean():.5f}, want {1 - th / 65536:.5f}")
    print("   corr along units: " + " ".join(f"lag{l} {c(k[:, :-l], k[:, l:]):+.5f}" for l in (1, 2, 3, 4, 8, 16, 32)) +
          f";  next pair {c(k[:-1], k[1:]):+.5f};  noise floor ~ {1 / np.sqrt(k.size):.5f}")
    print(f"   P(keep, keep) {float((keep[:, :-1] & keep[:, 1:]).mean()):.5f} (independent {keep.mean() ** 2:.5f}), "
          f"P(drop, drop) {float((~keep[:, :-1] & ~keep[:, 1:]).mean()):.5f} ({(1 - keep.mean()) ** 2:.5f})")
    print(f"   keep-count std per pair {keep.sum(1).std():.2f} (binomial {np.sqrt(nslab * 32 * p * (1 - p)):.2f}), per unit "
          f"{keep.sum(0).std():.2f} ({np.sqrt(P * p * (1 - p)):.2f})")


if __name__ == "__main__":
    main()
    k12()
