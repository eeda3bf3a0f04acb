#!/usr/bin/env python3
"""Micro-benchmarks of the individual HIP kernels at the BASELINE config-2 shapes (B=8 docs):
interleaved rounds in one process, HIP-event timing, median reported."""
import argparse
import math
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from peneo_amd import ops
from peneo_amd.hip import ACT_GELU


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="gemm,attn,pair,ln,dz")
    ap.add_argument("--B", type=int, default=8)
    args = ap.parse_args()
    dev = "cuda"
    dt = torch.bfloat16
    B, T, H, I, nh, d = args.B, 709, 768, 3072, 12, 64
    M = B * T
    what = args.what.split(",")
    if "gemm" in what:
        shapes = [("qkv  nt", M, 3 * H, H, True, True), ("ffn1 nt", M, I, H, True, True), ("ffn2 nt", M, H, I, True, True),
                  ("out  nt", M, H, H, True, True), ("dgrad nn ffn2", M, I, H, True, False), ("dgrad nn qkv", M, H, 3 * H, True, False),
                  ("wgrad tn ffn1", I, H, M, False, False), ("wgrad tn out", H, H, M, False, False),
                  ("z   nt chunk", 32768, 1920, 384, True, True), ("dx  nn chunk", 32768, 384, 1920, True, False),
                  ("dW1 tn chunk", 1920, 384, 32768, False, False), ("big 4096^3", 4096, 4096, 4096, True, True)]
        for name, m, n, k, ak, bk in shapes:
            a = torch.randn((m, k) if ak else (k, m), device=dev).to(dt)
            b = torch.randn((n, k) if bk else (k, n), device=dev).to(dt)
            out = torch.empty((m, n), device=dev, dtype=dt)
            ms = timeit(lambda: ops.gemm(a, b, a_kmajor=ak, b_kmajor=bk, out=out))
            print(f"gemm {name:16s} M={m:6d} N={n:5d} K={k:6d}: {ms * 1e3:8.1f} us  {2.0 * m * n * k / ms / 1e9:7.1f} TF/s  "
                  f"split_k={ops.choose_split_k(m, n, k, dt)}")
        bias = torch.randn(I, device=dev)
        a = torch.randn(M, H, device=dev).to(dt)
        b = torch.randn(I, H, device=dev).to(dt)
        pre = torch.empty(M, I, device=dev, dtype=dt)
        out = torch.empty(M, I, device=dev, dtype=dt)
        ms = timeit(lambda: ops.gemm(a, b, bias=bias, act=ACT_GELU, preact=pre, out=out))
        print(f"gemm ffn1+bias+gelu+preact: {ms * 1e3:8.1f} us  {2.0 * M * I * H / ms / 1e9:7.1f} TF/s")
    if "attn" in what:
        qkv = torch.randn(M, 3 * H, device=dev).to(dt)
        bias = torch.randn(B, nh, T, 768, device=dev).to(dt)
        bias[..., T:] = -1e30
        mask = None
        q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
        fl = 4.0 * T * T * d * nh * B
        ms = timeit(lambda: ops.attn_fwd(q, k, v, B, nh, T, d, 0.125, bias, mask))
        print(f"attn fwd (bias):   {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF/s")
        ms = timeit(lambda: ops.attn_fwd(q, k, v, B, nh, T, d, 0.125, None, None))
        print(f"attn fwd (nobias): {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF/s")
        out, lse = ops.attn_fwd(q, k, v, B, nh, T, d, 0.125, bias, mask)
        do = torch.randn_like(out)
        dqkv = torch.empty_like(qkv)
        g = torch.zeros(B, nh, T, 768, device=dev)
        ms = timeit(lambda: ops.attn_bwd(q, k, v, out, do, lse, B, nh, T, d, 0.125, bias, mask, dqkv, g))
        print(f"attn bwd (bias+G): {ms * 1e3:8.1f} us  {2.5 * fl / ms / 1e9:7.1f} TF/s (5 GEMM-equivalents)")
        ms = timeit(lambda: ops.attn_bwd(q, k, v, out, do, lse, B, nh, T, d, 0.125, None, None, dqkv, None))
        print(f"attn bwd (nobias): {ms * 1e3:8.1f} us  {2.5 * fl / ms / 1e9:7.1f} TF/s")
    if "pair" in what:
        N, D = 511, 384
        classes = [2, 3, 3, 3, 3]
        ab = torch.randn(B, N, 2 * D, device=dev).to(dt)
        w1 = [torch.randn(D, D, device=dev) / math.sqrt(D) for _ in classes]
        w2 = [torch.randn(c, D, device=dev) / math.sqrt(D) for c in classes]
        b1, b2 = torch.zeros(5 * D, device=dev), torch.zeros(14, device=dev)
        wp = ops.pair_heads_pack(dt, w1, w2)
        P = N * (N + 1) // 2
        fl = B * P * (2.0 * D * 5 * D + 2.0 * 5 * D * 14 / 5 * 5)
        ms = timeit(lambda: ops.pair_heads_fwd(ab, wp, b1, b2, classes))
        print(f"pair_heads_fwd (logits):      {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF/s")
        tags = [torch.zeros(B, P, dtype=torch.int64, device=dev) for _ in classes]
        cw = [torch.ones(c, device=dev) for c in classes]
        ms = timeit(lambda: ops.pair_heads_fwd(ab, wp, b1, b2, classes, tags=tags, class_weights=cw, want_dlogits=True))
        print(f"pair_heads_fwd (+CE+dlogits): {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF/s")
    if "ln" in what:
        x = torch.randn(M, H, device=dev).to(dt)
        g, bb = torch.ones(H, device=dev), torch.zeros(H, device=dev)
        y, mean, rstd = ops.layernorm_fwd(x, g, bb, 1e-5)
        ms = timeit(lambda: ops.layernorm_fwd(x, g, bb, 1e-5))
        print(f"ln fwd: {ms * 1e3:8.1f} us  {2 * x.numel() * 2 / ms / 1e6:7.1f} GB/s")
        dg, db = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
        ms = timeit(lambda: ops.layernorm_bwd(y, x, g, mean, rstd, dg, db))
        print(f"ln bwd: {ms * 1e3:8.1f} us  {3 * x.numel() * 2 / ms / 1e6:7.1f} GB/s")
        ms = timeit(lambda: ops.colsum(x))
        print(f"colsum: {ms * 1e3:8.1f} us  {x.numel() * 2 / ms / 1e6:7.1f} GB/s")
    if "dz" in what:
        npairs, D = 32768, 384
        classes = [2, 3, 3, 3, 3]
        z = torch.randn(npairs, 5 * D, device=dev).to(dt)
        w2 = [torch.randn(c, D, device=dev) for c in classes]
        dl = [torch.randn(npairs, c, device=dev) for c in classes]
        ws = ops.pair_dz_workspace(5, D, dev)
        sc = torch.ones(5, device=dev)
        ms = timeit(lambda: ops.pair_dz(z, npairs, D, classes, dl, w2, ws, sc))
        print(f"pair_dz: {ms * 1e3:8.1f} us  {2 * z.numel() * 2 / ms / 1e6:7.1f} GB/s")
        N = 511
        abd = torch.randn(N, 2 * D, device=dev).to(dt)
        i0, i1 = 0, 69
        npr = i1 * N - i1 * (i1 - 1) // 2
        dx = torch.randn(npr, D, device=dev).to(dt)
        dab = torch.zeros(N, 2 * D, device=dev)
        ms = timeit(lambda: ops.pair_x_bwd(abd, i0, i1, dx, dab))
        print(f"pair_x_bwd ({npr} pairs): {ms * 1e3:8.1f} us  {2 * dx.numel() * 2 / ms / 1e6:7.1f} GB/s (two passes over dx)")
        x = torch.empty(npr, D, device=dev, dtype=dt)
        ms = timeit(lambda: ops.pair_x_fwd(abd, i0, i1, x))
        print(f"pair_x_fwd: {ms * 1e3:8.1f} us  {x.numel() * 2 / ms / 1e6:7.1f} GB/s")


if __name__ == "__main__":
    main()
