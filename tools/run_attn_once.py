"""A few launches of the attention forward and the single-pass backward at the model's shape (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
B, nh, T, d = int(os.environ.get("B", "8")), 12, 709, 64
H = nh * d
dt = torch.bfloat16
drop = float(os.environ.get("DROP", "0.1"))
qkv = torch.randn(B * T, 3 * H, device="cuda").to(dt)
Tp = ops.attn_padded_len(T)
bias = (0.5 * torch.randn(B, nh, T, Tp, device="cuda")).to(dt)
q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
d_out = torch.randn(B * T, H, device="cuda").to(dt)
dqkv = torch.empty_like(qkv)
ds = torch.empty((B, nh, T, Tp), device="cuda", dtype=dt)
for _ in range(3):
    out, lse = ops.attn_fwd(q, k, v, B, nh, T, d, 0.125, bias, None, drop_p=drop, drop_seed=5)
    ops.attn_bwd(q, k, v, out, d_out, lse, B, nh, T, d, 0.125, bias, None, dqkv, None, drop_p=drop, drop_seed=5, ds_out=ds)
torch.cuda.synchronize()
