"""Fixed cost of the 128 x 128 GEMM kernel: one round of workgroups (504 tiles) against 1.58 rounds (810 tiles), K = 64 .. 768."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
from peneo_amd.hip import lib
lib().peneo_gemm_set_big_mode(0)
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
N = 2304
for M in (3584, 5672, 7168):
    row = []
    for K in (64, 128, 256, 768):
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        row.append(f"K={K}: {bench(lambda: ops.gemm(a, w, out=out)):5.1f}")
    tiles = ((M + 127) // 128) * (N // 128)
    print(f"M={M} ({tiles} tiles = {tiles / 512:.2f} rounds of 512)  " + "  ".join(row) + "  us")
