import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
DEV = "cuda"
B, N, D = 1, 511, 384
dtype, classes = torch.bfloat16, [2, 3, 3, 3, 3]
nh = len(classes)
g = torch.Generator().manual_seed(7)
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
outs = []
for rep in range(6):
    dz = torch.zeros((B * rows, nh * D), device=DEV, dtype=dtype)
    x = torch.zeros((B * rows, D), device=DEV, dtype=dtype)
    d_ab = torch.zeros(B, N, 2 * D, device=DEV)
    ws = ops.pair_dz_workspace(nh, D, DEV, slots=256)
    ops.pair_bwd_fused(ab, wp2, b1cat, args, dz, x, d_ab, ws)
    torch.cuda.synchronize()
    outs.append((dz.float(), d_ab.clone()))
print("non-repeatable elements vs rep 0: dz", [int(((o[0] - outs[0][0]).abs() > 0).sum()) for o in outs[1:]], " d_ab", [int(((o[1] - outs[0][1]).abs() > 0).sum()) for o in outs[1:]])
