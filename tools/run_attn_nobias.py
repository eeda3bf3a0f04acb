"""Attention forward / backward with and without the materialised bias tensor (how much of the time is the 104 MB bias stream?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
B, nh, T, d = int(os.environ.get("B", "8")), 12, 709, 64
H = nh * d
dt = torch.bfloat16
qkv = torch.randn(B * T, 3 * H, device="cuda").to(dt)
Tp = ops.attn_padded_len(T)
bias = (0.5 * torch.randn(B, nh, T, Tp, device="cuda")).to(dt)
bias[..., T:] = -1e30
kb = torch.zeros(B, Tp, device="cuda"); kb[:, T:] = -1e30
q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
d_out = torch.randn(B * T, H, device="cuda").to(dt)
dqkv = torch.empty_like(qkv)
ds = torch.empty((B, nh, T, Tp), dtype=dt, device="cuda")
def bench(name, fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1) / n * 1e3:8.1f} us")
for drop in (0.0, 0.1):
    w = ops.attn_drop_words(B, nh, T, drop, 5)[0] if drop > 0 else None
    for name, bb, kk in (("bias", bias, None), ("key_bias only", None, kb)):
        out, lse = ops.attn_fwd(q, k, v, B, nh, T, d, 0.125, bb, kk, drop_p=drop, drop_words=w)
        bench(f"fwd  p={drop} {name}", lambda: ops.attn_fwd(q, k, v, B, nh, T, d, 0.125, bb, kk, drop_p=drop, drop_words=w))
        bench(f"bwd  p={drop} {name}", lambda: ops.attn_bwd(q, k, v, out, d_out, lse, B, nh, T, d, 0.125, bb, kk, dqkv, None, drop_p=drop, drop_words=w, ds_out=ds))
