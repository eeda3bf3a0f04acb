"""Cold instruction cache: the four forward GEMMs of an encoder layer each in its own loop (the kernel stays in the instruction cache)
against the same four launched round-robin as a layer does (every launch starts from a cache that holds the previous kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
from peneo_amd.hip import ACT_GELU, ACT_NONE
M = 5672
cfg = (("QKV", 2304, 768, ACT_NONE, False), ("O", 768, 768, ACT_NONE, True), ("FFN1", 3072, 768, ACT_GELU, False), ("FFN2", 768, 3072, ACT_NONE, True))
calls = []
for name, N, K, act, res in cfg:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if res else None
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    calls.append((name, lambda a=a, w=w, bias=bias, act=act, r=r, out=out: ops.gemm(a, w, bias=bias, act=act, residual=r, out=out)))
def timed(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
alone = {name: timed(f, 40) for name, f in calls}
def layer():
    for _, f in calls: f()
rr = timed(layer, 40)
print("each in its own loop (us):", {k: round(v, 1) for k, v in alone.items()}, "sum", round(sum(alone.values()), 1))
print("round-robin, one 'layer' (us):", round(rr, 1), " -> cold-start penalty per layer", round(rr - sum(alone.values()), 1))
