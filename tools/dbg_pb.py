import math, sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, torch.nn.functional as F
from peneo_amd import ops
DEV = "cuda"
def rel_err(a, b): return float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-6))
for (B, N, D) in [(1, 70, 384), (1, 70, 384), (2, 16, 384), (1, 200, 384)]:
    dtype, classes = torch.bfloat16, [2, 3, 3, 3, 3]
    nh = len(classes)
    g = torch.Generator().manual_seed(B * 100000 + N * 1000 + D)
    ab = torch.randn(B, N, 2 * D, generator=g).to(DEV).to(dtype)
    P = N * (N + 1) // 2
    w1 = [(torch.randn(D, D, generator=g) / math.sqrt(D)).to(DEV) for _ in classes]
    w2 = [torch.randn(c, D, generator=g).to(DEV) for c in classes]
    b1cat = (0.1 * torch.randn(nh * D, generator=g)).to(DEV)
    dl = [torch.randn(B, P, c, generator=g).to(DEV) for c in classes]
    scale = torch.rand(nh, generator=g).to(DEV) + 0.5
    rows = ops.pair_bwd_rows(N)
    wp2 = ops.pair_bwd_pack(w1)
    args = ops.pair_dz_args(D, classes, dl, w2, scale)
    res = []
    for rep in range(2):
        dz = torch.zeros((B * rows, nh * D), device=DEV, dtype=dtype)
        x = torch.zeros((B * rows, D), device=DEV, dtype=dtype)
        d_ab = torch.zeros(B, N, 2 * D, device=DEV)
        ws = ops.pair_dz_workspace(nh, D, DEV, slots=256)
        ops.pair_bwd_fused(ab, wp2, b1cat, args, dz, x, d_ab, ws)
        torch.cuda.synchronize()
        res.append((dz.clone(), x.clone(), d_ab.clone()))
    print(B, N, D, "repeatable dz:", torch.equal(res[0][0], res[1][0]), "x:", torch.equal(res[0][1], res[1][1]), "d_ab:", torch.equal(res[0][2], res[1][2]))
    dz = res[0][0]
    # reference dz in block order is awkward; compare column norms per slab against the old kernel path instead
    wp = ops.pair_heads_pack(dtype, w1, w2)
    tot = torch.zeros(nh * D, device=DEV)
    for b in range(B):
        zb = torch.empty(P, nh * D, device=DEV, dtype=dtype)
        a2 = ops.pair_dz_args(D, classes, [d[b] for d in dl], w2, scale)
        ops.pair_dz_fused(ab[b], 0, N, wp, b1cat, a2, zb, ops.pair_dz_workspace(nh, D, DEV))
        tot += zb.float().pow(2).sum(0)
    mine = dz.float().pow(2).sum(0)
    r = (mine - tot).abs() / tot.clamp_min(1e-9)
    bad = torch.nonzero(r > 1e-2).flatten()
    print("  columns whose energy differs >1%:", bad.numel(), bad[:20].tolist(), "slabs:", sorted(set((bad // 32).tolist()))[:20])
