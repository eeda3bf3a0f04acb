"""GEMM throughput on the model's shapes (B = 8 docs: R = 5672 rows), all three layouts; prints TF/s per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops

R = 5672
dt = torch.bfloat16
def bench(name, fn, flops, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:34s} {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TF/s")

for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    x = torch.randn(R, K, device="cuda").to(dt)
    w = torch.randn(N, K, device="cuda").to(dt)
    dy = torch.randn(R, N, device="cuda").to(dt)
    fl = 2.0 * R * N * K
    bench(f"fwd   x[{R},{K}] W[{N},{K}]^T", lambda: ops.gemm(x, w), fl)
    bench(f"dgrad dy[{R},{N}] W[{N},{K}]", lambda: ops.gemm(dy, w, b_kmajor=False), fl)
    bench(f"wgrad dy^T x -> [{N},{K}] fp32", lambda: ops.gemm(dy, x, a_kmajor=False, b_kmajor=False, out_dtype=torch.float32), fl)
a = torch.randn(4096, 4096, device="cuda").to(dt)
bench("4096^3 TN", lambda: ops.gemm(a, a), 2.0 * 4096 ** 3)
bench("4096^3 NN", lambda: ops.gemm(a, a, a_kmajor=False, b_kmajor=False), 2.0 * 4096 ** 3)
