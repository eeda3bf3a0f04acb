import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
from peneo_amd.hip import ACT_GELU
R, H, I = 5672, 768, 3072
dt = torch.bfloat16
x = torch.randn(R, H, device="cuda").to(dt); w = (torch.randn(I, H, device="cuda") * 0.03).to(dt); b = torch.zeros(I, device="cuda")
z = torch.empty(R, I, device="cuda", dtype=dt); dy = torch.randn(R, H, device="cuda").to(dt); w2 = (torch.randn(H, I, device="cuda") * 0.03).to(dt)
def bench(name, fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1) / n * 1e3:8.1f} us")
bench("FFN1 fwd plain", lambda: ops.gemm(x, w, bias=b))
bench("FFN1 fwd + GELU + preact", lambda: ops.gemm(x, w, bias=b, act=ACT_GELU, preact=z))
bench("FFN2 dgrad plain", lambda: ops.gemm(dy, w2, b_kmajor=False))
bench("FFN2 dgrad * GELU'(z)", lambda: ops.gemm(dy, w2, b_kmajor=False, grad_src=z, grad_act=ACT_GELU))
out = ops.gemm(x, w, bias=b, act=ACT_GELU)
ref = torch.nn.functional.gelu(x.float() @ w.float().t())
print("gelu max abs err vs torch:", float((out.float() - ref).abs().max()), "ref max", float(ref.abs().max()))
