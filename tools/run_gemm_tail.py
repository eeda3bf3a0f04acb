"""Is the 128 x 128 kernel's time at M = 5672, N = 768 the tail of its 270 tiles on 256 CUs?  The same GEMM at row counts that give 192 .. 300 tiles
(K = 3072: FFN2 forward / FFN1 dgrad; K = 768: out-projection): a step at 256 tiles is the double-loaded CUs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
from peneo_amd.hip import lib
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
N = 768
for mode in (0, 1):
    lib().peneo_gemm_set_big_mode(mode)
    print("big-tile mode", mode, "(0 = the 128 x 128 kernel only, 1 = the cost model's choice)")
    for K in (3072, 768):
        row = []
        for M in (4096, 5120, 5376, 5461, 5504, 5672, 6144, 6400):
            a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
            w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            t = bench(lambda: ops.gemm(a, w, out=out))
            tiles = ((M + 127) // 128) * (N // 128)
            row.append(f"M={M} ({tiles} tiles) {t:5.1f} us = {t / M * 1e3:5.2f} ns/row")
        print(f"  K={K}: " + " | ".join(row))
lib().peneo_gemm_set_big_mode(1)
