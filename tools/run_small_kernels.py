"""Bandwidth of the memory-bound helpers at the model's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
R, H = 5672, 768
dt = torch.bfloat16
def bench(name, fn, nbytes, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:30s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e9:7.2f} TB/s")
x = torch.randn(R, H, device="cuda").to(dt); dy = torch.randn(R, H, device="cuda").to(dt)
g, b = torch.ones(H, device="cuda"), torch.zeros(H, device="cuda")
y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5)
dg, db = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
bench("ln_fwd", lambda: ops.layernorm_fwd(x, g, b, 1e-5), 2 * R * H * 2)
bench("ln_fwd drop", lambda: ops.layernorm_fwd(x, g, b, 1e-5, drop_p=0.1, drop_seed=3), 2 * R * H * 2)
bench("ln_bwd", lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dg, db), 3 * R * H * 2)
bench("ln_bwd drop(dy)", lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dg, db, drop_p=0.1, drop_seed=3), 3 * R * H * 2)
dxd = torch.empty_like(x)
bench("ln_bwd + dropped 2nd output", lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dx_dropped=dxd, drop2_p=0.1, drop2_seed=4), 4 * R * H * 2)
big = torch.randn(8, 709, H, device="cuda").to(dt)
bench("ln_bwd drop, sliced [B,512,H] of [B,709,H]", lambda: ops.layernorm_bwd(big[:, :512], big[:, :512], g, mean[:4096], rstd[:4096], dg, db, dx=torch.empty(8, 512, H, device="cuda", dtype=dt), drop_p=0.1, drop_seed=3), 3 * 4096 * H * 2)
for N in (768, 2304, 3072):
    z = torch.randn(R, N, device="cuda").to(dt)
    bench(f"colsum N={N}", lambda: ops.colsum(z), R * N * 2)
