"""Where the waves of attn_bwd_fused_kernel spend their cycles (s_memtime, -DATTN_PROF build: bash tools/prof_build.sh attn, then run with PENEO_HIP_LIB=$PWD/peneo_amd/lib/libpeneo_attnprof.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops, hip
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
out, lse = ops.attn_fwd(q, k, v, B, nh, T, d, 0.125, bias, None, drop_p=drop, drop_seed=5)
lib = hip.lib()
buf = (C.c_ulonglong * 16)()
for _ in range(2):
    ops.attn_bwd(q, k, v, out, d_out, lse, B, nh, T, d, 0.125, bias, None, dqkv, None, drop_p=drop, drop_seed=5, ds_out=ds)
torch.cuda.synchronize()
lib.peneo_attn_prof_read(buf, 1)
n = 5
for _ in range(n):
    ops.attn_bwd(q, k, v, out, d_out, lse, B, nh, T, d, 0.125, bias, None, dqkv, None, drop_p=drop, drop_seed=5, ds_out=ds)
torch.cuda.synchronize()
lib.peneo_attn_prof_read(buf, 1)
v_ = [float(x) for x in buf]
waves = v_[11]
names = ["wait at barrier A (others still in the previous tile)", "Q / dO / bias tile stores to LDS (incl. the wait for their global loads)",
         "wait at barrier B", "issue the next tile's global loads", "S and dP: 8 fragment reads + 8 MFMAs per 32 queries",
         "softmax backward (4 groups: lse / delta / bias reads, exp, dS, LDS store)", "dV / dK: 16 transpose-read pairs + 8 MFMAs per 32 queries",
         "wait at barrier C", "dS^T tile LDS -> HBM", "loop / branch overhead in front of the barriers"]
tot = v_[10]
print(f"B = {B}: {int(waves / n)} waves per launch, {tot / waves:9.0f} ticks per wave in the tile loop ({tot / waves / 12:7.0f} per 64-query tile)")
for i, nm in enumerate(names):
    print(f"  {100 * v_[i] / tot:5.1f} %  {v_[i] / waves / 12:7.0f} ticks / tile   {nm}")
