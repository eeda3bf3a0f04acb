"""Debug: dz of peneo_pair_bwd_fused with the classifier dropout against (dz without dropout) x host mask x scale."""
import math, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from peneo_amd import ops
from dropout_ref import k12_keep, k12_scale
DEV = "cuda"
B, N, D, p_drop, seed = 1, 45, 128, 0.25, 4242
classes = [2, 3, 3, 3, 3]; nh = 5
g = torch.Generator().manual_seed(1)
ab = torch.randn(B, N, 2 * D, generator=g).to(DEV).to(torch.bfloat16)
P = N * (N + 1) // 2
w1 = [(torch.randn(D, D, generator=g) / math.sqrt(D)).to(DEV) for _ in classes]
w2 = [torch.randn(c, D, generator=g).to(DEV) for c in classes]
b1cat = (0.1 * torch.randn(nh * D, generator=g)).to(DEV)
dl = [torch.randn(B, P, c, generator=g).to(DEV) for c in classes]
scale = torch.ones(nh, device=DEV)
rows = ops.pair_bwd_rows(N)
wp2 = ops.pair_bwd_pack(w1)
def run(p):
    args = ops.pair_dz_args(D, classes, dl, w2, scale, drop_p=p, drop_seed=seed)
    dz = torch.zeros((B * rows, nh * D), device=DEV, dtype=torch.bfloat16)
    x = torch.zeros((B * rows, D), device=DEV, dtype=torch.bfloat16)
    d_ab = torch.zeros((B, N, 2 * D), device=DEV)
    ws = ops.pair_dz_workspace(nh, D, DEV, slots=256)
    ops.pair_bwd_fused(ab, wp2, b1cat, args, dz, x, d_ab, ws)
    torch.cuda.synchronize()
    return dz.float().cpu()
dz0, dz1 = run(0.0), run(p_drop)
keep = k12_keep(seed, 0, 0, P, nh * D, p_drop)
# block order -> pair index
nti, ntj = (N + 7) // 8, (N + 15) // 16
pidx = torch.full((rows,), -1, dtype=torch.long)
blk = 0
for ti in range(nti):
    for tj in range(ti >> 1, ntj):
        for grp in range(4):
            for r in range(32):
                pi, pj = ti * 8 + 2 * grp + (r >> 4), tj * 16 + (r & 15)
                if pi < N and pj < N and pi <= pj:
                    pidx[blk * 128 + grp * 32 + r] = pi * N - pi * (pi - 1) // 2 + (pj - pi)
        blk += 1
ok = pidx >= 0
want = dz0[ok] * keep[pidx[ok]] * k12_scale(p_drop)
got = dz1[ok]
bad = (got - want).abs() > 0.02 * want.abs().max()
print("mismatching elements:", int(bad.sum()), "of", bad.numel(), " keep rate got", float((got != 0).float().mean()), "want", float((want != 0).float().mean()))
# pattern: by hidden column within slab, by row position in group
rr = torch.nonzero(ok).squeeze(1)
rowpos = (rr % 32)
bycol = bad.view(-1, nh * D // 32, 32).float().mean((0, 1))
print("by hidden column in slab:", [round(float(v), 2) for v in bycol])
byrow = torch.zeros(32); cnt = torch.zeros(32)
byrow.index_add_(0, rowpos, bad.float().mean(1)); cnt.index_add_(0, rowpos, torch.ones(len(rowpos)))
print("by pair position in group:", [round(float(v), 2) for v in (byrow / cnt.clamp_min(1))])
byslab = bad.view(bad.shape[0], -1, 32).float().mean((0, 2))
print("by slab:", [round(float(v), 2) for v in byslab])
# does got match a mask at all? compare zero patterns
z_got, z_want = (got == 0), (want == 0)
print("zero-pattern agreement:", float((z_got == z_want).float().mean()))
