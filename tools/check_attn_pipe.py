"""The pipelined attention backward (attn_bwd_pipe.hip) against the fused kernel it replaces: bit-for-bit dq | dk | dv and dS^T slab
(the old kernel is selected by also asking for the fp32 bias gradient, which the new one does not produce), then timings at the
model's shape.  python tools/check_attn_pipe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
dt = torch.bfloat16
bad = 0
for (B, nh, T, d, drop) in [(1, 2, 709, 64, 0.0), (2, 3, 709, 64, 0.1), (2, 2, 200, 64, 0.2), (1, 1, 64, 64, 0.1), (2, 2, 33, 64, 0.0),
                            (1, 16, 1221, 64, 0.1), (3, 2, 128, 64, 0.1), (1, 2, 129, 64, 0.1), (20, 16, 140, 64, 0.1), (20, 10, 300, 64, 0.0),
                            (8, 12, 709, 64, 0.1)]:
    g = torch.Generator().manual_seed(T + int(drop * 100))
    H = nh * d
    qkv = torch.randn(B * T, 3 * H, generator=g).to("cuda").to(dt)
    Tp = ops.attn_padded_len(T)
    bias = torch.full((B, nh, T, Tp), -1.0e30, dtype=dt, device="cuda")
    bias[..., :T] = (0.5 * torch.randn(B, nh, T, T, generator=g)).to("cuda").to(dt)
    bias[0, :, :, T // 3: T // 2] = -1.0e30
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    out, lse = ops.attn_fwd(q, k, v, B, nh, T, d, 0.125, bias, None, drop_p=drop, drop_seed=5)
    d_out = torch.randn(B * T, H, generator=g).to("cuda").to(dt)
    r = []
    for use_old in (True, False):
        dqkv = torch.full_like(qkv, 3.0)
        ds = torch.full((B, nh, T, Tp), 7.0, device="cuda", dtype=dt)
        gb = torch.zeros(bias.shape, dtype=torch.float32, device="cuda") if use_old else None
        ops.attn_bwd(q, k, v, out, d_out, lse, B, nh, T, d, 0.125, bias, None, dqkv, gb, drop_p=drop, drop_seed=5, ds_out=ds)
        torch.cuda.synchronize()
        r.append((dqkv.float(), ds.float()))
    names = ["dq", "dk", "dv"]
    line = f"B={B} nh={nh} T={T} drop={drop}:"
    for i, n in enumerate(names):
        a, b_ = r[0][0][:, i * H:(i + 1) * H], r[1][0][:, i * H:(i + 1) * H]
        nd = int((a != b_).sum())
        rel = float((a - b_).norm() / (a.norm() + 1e-30))
        line += f" {n} differ {nd} rel {rel:.2e};"
        bad += rel > 1e-3 or not torch.isfinite(b_).all()
    a, b_ = r[0][1], r[1][1]
    nd = int((a != b_).sum())
    rel = float((a - b_).norm() / (a.norm() + 1e-30))
    line += f" dS^T differ {nd} rel {rel:.2e}; pad max {float(b_[..., T:].abs().max()) if Tp > T else 0.0}"
    bad += rel > 1e-3 or (Tp > T and float(b_[..., T:].abs().max()) != 0.0)
    print(line, flush=True)
print("FAILED" if bad else "all close")

B, nh, T, d = int(os.environ.get("B", "8")), 12, 709, 64
H = nh * d
for drop in (0.1, 0.0):
    qkv = torch.randn(B * T, 3 * H, device="cuda").to(dt)
    Tp = ops.attn_padded_len(T)
    bias = (0.5 * torch.randn(B, nh, T, Tp, device="cuda")).to(dt)
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    w = ops.attn_drop_words(B, nh, T, drop, 5)[0] if drop > 0 else None
    out, lse = ops.attn_fwd(q, k, v, B, nh, T, d, 0.125, bias, None, drop_p=drop, drop_words=w)
    d_out = torch.randn(B * T, H, device="cuda").to(dt)
    dqkv = torch.empty_like(qkv)
    ds = torch.empty((B, nh, T, Tp), device="cuda", dtype=dt)
    gb = torch.zeros(bias.shape, dtype=torch.float32, device="cuda")
    def bench(name, fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"drop={drop} {name:40s} {e0.elapsed_time(e1) / n * 1e3:8.1f} us", flush=True)
    bench("delta + pipe + dQ-from-dS (new)", lambda: ops.attn_bwd(q, k, v, out, d_out, lse, B, nh, T, d, 0.125, bias, None, dqkv, None, drop_p=drop, drop_words=w, ds_out=ds))
    bench("delta + fused + dQ-from-dS + G atomics (old)", lambda: ops.attn_bwd(q, k, v, out, d_out, lse, B, nh, T, d, 0.125, bias, None, dqkv, gb, drop_p=drop, drop_words=w, ds_out=ds))
