#!/usr/bin/env python3
"""Statistics of the two counter-based dropout generators, in numpy (CPU only).

attention (attention.hip: attn_drop_words_kernel): one dword = the 32 queries of a block for one key, a chain of 32 fields
(st <- st[23:0] * 0xC2B2AF + 0x9E3779, field = st >> 16) seeded by mix32(seed ^ word index * 0x9E3779B9).  classifier (common.h:
pair_drop_*): one chain of 16 fields per (pair, slab, half), seeded by a 24-bit mix.  Prints keep rates, correlations along
the chains and across neighbouring chains, joint probabilities and the spread of keep counts against the binomial expectation."""
import numpy as np

M = np.uint64(0xFFFFFFFF)
u = np.uint64


def mix32(x):
    x = x.astype(np.uint64)
    x ^= x >> u(16); x = (x * u(0x7FEB352D)) & M; x ^= x >> u(15); x = (x * u(0x846CA68B)) & M; x ^= x >> u(16)
    return x


def premix(x, premul=True):
    x = x.astype(np.uint64)
    if premul:
        x = (x * u(0x9E3779B1)) & M
    return x ^ (x >> u(16))


def mix24(x, premul=True):
    """pair_drop_seed's mixer; premul=False is the round-3/4 form without the full 32-bit multiply in front."""
    x = premix(x, premul); x = ((x & u(0xFFFFFF)) * u(0x9E3779)) & M
    x ^= x >> u(13); x = ((x & u(0xFFFFFF)) * u(0x85EBCB)) & M
    x ^= x >> u(16)
    return x


def main(p=0.1, nqb=24, Tk=768, bh=8, seed=12345):
    """Attention keep words: [bh][query block][key slot] dwords, bit i = query 32 qb + i."""
    n = bh * nqb * Tk
    idx = np.arange(n, dtype=np.uint64)
    st = mix32(u(seed) ^ ((idx * u(0x9E3779B9)) & M))
    th = round(p * 65536)
    keep = np.zeros((n, 32), dtype=bool)
    inc = u(0x9E3779) ^ ((st >> u(24)) << u(8))
    for bit in range(32):
        st = (((st & u(0xFFFFFF)) * u(0xC2B2AF)) + inc) & M
        keep[:, bit] = (st >> u(16)) >= th
    # [bh, q = 32 qb + bit, key slot]
    m = keep.reshape(bh, nqb, Tk, 32).transpose(0, 1, 3, 2).reshape(bh, nqb * 32, Tk)
    k = m.astype(float) - m.mean()
    c = lambda a, b: float((a * b).mean() / k.var())
    print(f"attention words: keep {m.mean():.5f}, want {1 - th / 65536:.5f}")
    print("   corr along queries (the chain): " + " ".join(f"lag{l} {c(k[:, :-l], k[:, l:]):+.5f}" for l in (1, 2, 3, 4, 8, 31, 32)) +
          ";  along key slots: " + " ".join(f"lag{l} {c(k[:, :, :-l], k[:, :, l:]):+.5f}" for l in (1, 2, 64)) +
          f";  next (b, h) {c(k[:-1], k[1:]):+.5f};  noise floor ~ {1 / np.sqrt(k.size):.5f}")
    print(f"   P(keep, keep) along queries {float((m[:, :-1] & m[:, 1:]).mean()):.5f} (independent {m.mean() ** 2:.5f}), "
          f"P(drop, drop) {float((~m[:, :-1] & ~m[:, 1:]).mean()):.5f} ({(1 - m.mean()) ** 2:.5f})")
    print(f"   keep-count std per query row {m.sum(2).std():.2f} (binomial {np.sqrt(Tk * p * (1 - p)):.2f}), per key "
          f"{m.sum(1).std():.2f} ({np.sqrt(nqb * 32 * p * (1 - p)):.2f})")


def k12(p=0.1, P=3000, nslab=60, seed=12345):
    """The classifier-dropout mask (common.h: pair_drop_seed / pair_drop_step): one mix24 seed per (pair, slab, half), 16
    fields by a 24-bit multiply-add chain.  Keep rate, correlations along the chain (lag 1..3), across halves (lag 4),
    groups (lag 8), slabs (lag 32) and pairs, joint probabilities of neighbours, count statistics."""
    key = int(mix32(np.array([(seed ^ 0x9E3779B9) & 0xFFFFFFFF], dtype=np.uint64))[0])
    pp = np.arange(P, dtype=np.uint64)[:, None, None]; sl = np.arange(nslab, dtype=np.uint64)[None, :, None]
    hh = np.arange(2, dtype=np.uint64)[None, None, :]
    cnt = (((pp * u(nslab) + sl) * u(2) + hh) & M) ^ u(key)
    st = mix24(cnt)
    f = np.zeros((P, nslab, 32), dtype=np.uint64)
    inc = u(0x9E3779) ^ ((premix(cnt) >> u(24)) << u(8))
    for i in range(16):
        st = (((st & u(0xFFFFFF)) * u(0xC2B2AF)) + inc) & M
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


def k12_far_duplicates(n=83_800_000, seed=12345, chunk=1 << 22):
    """Chains (counters 0 .. n - 1: config 4 has N = 1023, P = 523 776 pairs x 80 slabs x 2 halves = 83.8 M) that carry the same
    16-element mask as another chain.  A chain is determined by the low 24 bits of its seed word (the LCG state) and, since
    round 5, by bits 24..31 (the LCG increment): `bits` = 24 counts duplicates of the round-3/4 chain, 32 of the current one.  A random map of n counters into 2^32 states leaves
    about n^2 / 2^33 colliding pairs (birthday level: 8.2e5 at n = 83.8 M, i.e. 1.9 % of the chains); the round-3/4 seed paired
    every counter above 2^24 with the one at c ^ 0x01000100 (~80 % of the chains)."""
    key = int(mix32(np.array([(seed ^ 0x9E3779B9) & 0xFFFFFFFF], dtype=np.uint64))[0])
    for premul, bits in ((False, 24), (True, 24), (True, 32)):
        st = np.empty(n, dtype=np.uint32)
        for a in range(0, n, chunk):
            c = np.arange(a, min(n, a + chunk), dtype=np.uint64)
            w = mix24(c ^ u(key), premul) & u(0xFFFFFF)          # the LCG state: 24 bits
            if bits == 32:
                w = w | ((premix(c ^ u(key), premul) >> u(24)) << u(24))   # + the 8 premixed bits that select the increment
            st[a:a + len(c)] = w.astype(np.uint32)
        partner = np.arange(n, dtype=np.int64) ^ 0x01000100
        ok = partner < n
        structural = int((st[ok] == st[partner[ok]]).sum())
        st.sort()
        dup = int((st[1:] == st[:-1]).sum())
        print(f"K12 seeds, {n / 1e6:.1f} M chains, {'with' if premul else 'without'} the 32-bit pre-multiply, {bits} seed bits select the chain: "
              f"chains whose partner at c ^ 0x01000100 carries the same mask: {structural} ({structural / n:.1%}); duplicates in all (equal "
              f"neighbours after sorting): {dup} (birthday expectation for 2^{bits} chains: {n * n / 2 ** (bits + 1):.3g})")


if __name__ == "__main__":
    import sys
    main()
    k12()
    if "--far" in sys.argv:
        k12_far_duplicates()
