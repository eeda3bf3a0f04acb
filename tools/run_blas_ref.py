"""Headroom check: the vendor library (torch.mm -> hipBLASLt/rocBLAS) against peneo_gemm on the decoder / encoder shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
dt = torch.bfloat16
def bench(name, fn, flops, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:44s} {ms * 1e3:9.1f} us  {flops / ms / 1e9:7.1f} TF/s")
P, D, H5 = 130816, 384, 1920
x = torch.randn(P, D, device="cuda").to(dt); w1 = torch.randn(H5, D, device="cuda").to(dt); dz = torch.randn(P, H5, device="cuda").to(dt)
fl = 2.0 * P * D * H5
z = torch.empty(P, H5, device="cuda", dtype=dt); dx = torch.empty(P, D, device="cuda", dtype=dt); dW = torch.zeros(H5, D, device="cuda")
bench("z  = x W1^T        peneo", lambda: ops.gemm(x, w1, out=z), fl)
bench("z  = x W1^T        torch", lambda: torch.mm(x, w1.t(), out=z), fl)
bench("dx = dz W1         peneo", lambda: ops.gemm(dz, w1, b_kmajor=False, out=dx), fl)
bench("dx = dz W1         torch", lambda: torch.mm(dz, w1, out=dx), fl)
bench("dW = dz^T x (fp32) peneo", lambda: ops.gemm(dz, x, a_kmajor=False, b_kmajor=False, out=dW, accumulate=True), fl)
dWb = torch.empty(H5, D, device="cuda", dtype=dt)
bench("dW = dz^T x (bf16) torch", lambda: torch.mm(dz.t(), x, out=dWb), fl)
R = 5672
for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    a = torch.randn(R, K, device="cuda").to(dt); w = torch.randn(N, K, device="cuda").to(dt); dy = torch.randn(R, N, device="cuda").to(dt)
    f2 = 2.0 * R * N * K
    bench(f"fwd [{R},{K}]x[{N},{K}]^T peneo", lambda: ops.gemm(a, w), f2)
    bench(f"fwd [{R},{K}]x[{N},{K}]^T torch", lambda: torch.mm(a, w.t()), f2)
    bench(f"wgrad -> [{N},{K}] peneo", lambda: ops.gemm(dy, a, a_kmajor=False, b_kmajor=False, out_dtype=torch.float32), f2)
    bench(f"wgrad -> [{N},{K}] torch", lambda: torch.mm(dy.t(), a), f2)
    # dgrad of the same layer: dx [R, K] = dy [R, N] . W [N, K]  (B mn-major)
    bench(f"dgrad [{R},{N}]x[{N},{K}] peneo", lambda: ops.gemm(dy, w, b_kmajor=False), f2)
    bench(f"dgrad [{R},{N}]x[{N},{K}] torch", lambda: torch.mm(dy, w), f2)
