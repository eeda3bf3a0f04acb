"""Fixed cost (prologue + epilogue) against the per-k-tile cost of the forward GEMM kernels: time over K at M = 5672."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
def bench(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 5672
for N in (2304, 3072, 768):
    row = []
    for K in (128, 256, 512, 768, 1536, 3072):
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        t0 = bench(lambda: ops.gemm(a, w, bias=bias))
        t2 = bench(lambda: torch.nn.functional.linear(a, w, bias.to(torch.bfloat16)))
        row.append(f"K={K}: {t0:5.1f}/{t2:5.1f}")
    print(f"N={N}  (peneo_gemm / torch, us)  " + "  ".join(row))
