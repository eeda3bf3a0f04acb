import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
DEV = "cuda"
for (B, N, D) in [(1, 200, 384), (1, 511, 384), (2, 130, 384)]:
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
    for rep in range(4):
        dz = torch.zeros((B * rows, nh * D), device=DEV, dtype=dtype)
        x = torch.zeros((B * rows, D), device=DEV, dtype=dtype)
        d_ab = torch.zeros(B, N, 2 * D, device=DEV)
        ws = ops.pair_dz_workspace(nh, D, DEV, slots=256)
        ops.pair_bwd_fused(ab, wp2, b1cat, args, dz, x, d_ab, ws)
        torch.cuda.synchronize()
        res.append((dz.float().clone(), d_ab.clone()))
    for rep in range(1, 4):
        diff = (res[rep][0] - res[0][0]).abs()
        nzr = torch.nonzero(diff.sum(1)).flatten()
        nzc = torch.nonzero(diff.sum(0)).flatten()
        print(B, N, "rep", rep, "dz max diff", float(diff.max()), "rows differing", nzr.numel(), nzr[:8].tolist(), "tile", (nzr[:8] // 128).tolist(),
              "cols", nzc.numel(), nzc[:8].tolist(), "| d_ab diff", float((res[rep][1] - res[0][1]).abs().max()))
