"""One whole-document launch of pair_dz_fused (for the PMC traffic passes of tools/pmc_traffic.sh)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
N, D, classes, nh = 511, 384, [2, 3, 3, 3, 3], 5
dt, dev = torch.bfloat16, "cuda"
ab = torch.randn(N, 2 * D, device=dev).to(dt)
P = N * (N + 1) // 2
w1 = [(torch.randn(D, D, device=dev) / math.sqrt(D)) for _ in classes]
w2 = [torch.randn(c, D, device=dev) for c in classes]
b1 = torch.zeros(nh * D, device=dev)
dl = [torch.randn(P, c, device=dev) for c in classes]
scale = torch.ones(nh, device=dev)
wp = ops.pair_heads_pack(dt, w1, w2)
z = torch.empty(P, nh * D, device=dev, dtype=dt)
ws = ops.pair_dz_workspace(nh, D, dev, slots=256)
args = ops.pair_dz_args(D, classes, dl, w2, scale)
for _ in range(3):
    ops.pair_dz_fused(ab, 0, N, wp, b1, args, z, ws)
torch.cuda.synchronize()
print("algorithmic bytes: dz", P * nh * D * 2 / 1e6, "MB written; ab", ab.numel() * 2 / 1e6, "+ dlogits", sum(t.numel() for t in dl) * 4 / 1e6, "MB read")
