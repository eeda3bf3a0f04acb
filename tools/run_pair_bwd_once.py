"""Three launches of peneo_pair_bwd_fused at config 2 (B = 8, N = 511, D = 384) — the launch bench.py times in the train
step; for the PMC traffic passes of tools/pmc_traffic.sh (kernel substring: pair_bwd_ws_kernel)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
B, N, D, classes, nh = 8, 511, 384, [2, 3, 3, 3, 3], 5
dt, dev = torch.bfloat16, "cuda"
ab = torch.randn(B, N, 2 * D, device=dev).to(dt)
P = N * (N + 1) // 2
w1 = [(torch.randn(D, D, device=dev) / math.sqrt(D)) for _ in classes]
w2 = [torch.randn(c, D, device=dev) for c in classes]
b1 = torch.zeros(nh * D, device=dev)
dl = [torch.randn(B, P, c, device=dev) * 1e-3 for c in classes]
scale = torch.ones(nh, device=dev)
wp2 = ops.pair_bwd_pack(w1)
rows = ops.pair_bwd_rows(N)
dz = torch.empty(B * rows, nh * D, device=dev, dtype=dt)
x = torch.empty(B * rows, D, device=dev, dtype=dt)
d_ab = torch.zeros(B, N, 2 * D, device=dev)
ws = ops.pair_dz_workspace(nh, D, dev, slots=256)
args = ops.pair_dz_args(D, classes, dl, w2, scale, drop_p=0.1, drop_seed=1234)   # train step: the forward's classifier dropout regenerated
for _ in range(3):
    ops.pair_bwd_fused(ab, wp2, b1, args, dz, x, d_ab, ws)
torch.cuda.synchronize()
print("algorithmic bytes per launch: dz", B * P * nh * D * 2 / 1e6, "+ x", B * P * D * 2 / 1e6, "MB written (the kernel writes", B * rows * (nh + 1) * D * 2 / 1e6,
      "MB in block order); ab", ab.numel() * 2 / 1e6, "+ dlogits", sum(t.numel() for t in dl) * 4 / 1e6, "MB read; partial rows", B * (rows // 128) * 24 * D * 4 / 1e6, "MB")
