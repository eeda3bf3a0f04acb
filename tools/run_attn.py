"""Attention forward / backward timing at the model's shape (B=8, nh=12, T=709, d=64, bf16, with bias)."""
import math, os, sys
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
def bench(name, fn, flops, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:32s} {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TF/s")
fl = 4.0 * B * nh * T * T * d
w = ops.attn_drop_words(B, nh, T, drop, 5)[0] if drop > 0 else None
if drop > 0:
    bench('drop words, 1 layer', lambda: ops.attn_drop_words(B, nh, T, drop, 5), fl)
    bench('drop words, 12 layers', lambda: ops.attn_drop_words(B, nh, T, drop, 5, sets=12), fl)
out, lse = ops.attn_fwd(q, k, v, B, nh, T, d, 0.125, bias, None, drop_p=drop, drop_words=w)
bench("attn_fwd", lambda: ops.attn_fwd(q, k, v, B, nh, T, d, 0.125, bias, None, drop_p=drop, drop_words=w), fl)
d_out = torch.randn(B * T, H, device="cuda").to(dt)
dqkv = torch.empty_like(qkv)
g = torch.zeros(bias.shape, dtype=torch.float32, device="cuda")
bench("attn_bwd single pass (+G)", lambda: ops.attn_bwd(q, k, v, out, d_out, lse, B, nh, T, d, 0.125, bias, None, dqkv, g, drop_p=drop, drop_words=w), 2.5 * fl)
bench("attn_bwd single pass, dQ atomics (no G)", lambda: ops.attn_bwd(q, k, v, out, d_out, lse, B, nh, T, d, 0.125, bias, None, dqkv, None, drop_p=drop, drop_words=w, dq_atomic=True), 2.5 * fl)
bench("attn_bwd single pass (no G)", lambda: ops.attn_bwd(q, k, v, out, d_out, lse, B, nh, T, d, 0.125, bias, None, dqkv, None, drop_p=drop, drop_words=w), 2.5 * fl)
bench("attn_bwd two kernels (+G)", lambda: ops.attn_bwd(q, k, v, out, d_out, lse, B, nh, T, d, 0.125, bias, None, dqkv, g, drop_p=drop, drop_words=w, single_pass=False), 2.5 * fl)
