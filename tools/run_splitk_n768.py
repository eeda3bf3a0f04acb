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
    print(f"{name:44s} {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TF/s")
for K in (768, 2304, 3072):
    x = torch.randn(R, K, device="cuda").to(dt); w = torch.randn(768, K, device="cuda").to(dt)
    b = torch.zeros(768, device="cuda"); res = torch.randn(R, 768, device="cuda").to(dt)
    fl = 2.0 * R * 768 * K
    for sk in (1, 2, 3):
        bench(f"[5672,{K}]x[768,{K}]^T +bias+res split_k={sk}", lambda: ops.gemm(x, w, bias=b, residual=res, split_k=sk), fl)
