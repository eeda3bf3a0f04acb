"""The saving forward and peneo_pair_bwd_saved at config 2 (B = 8, N = 511, D = 384), three launches each - the two launches bench.py times in
the train step since round 5; for the PMC traffic passes of tools/pmc_traffic.sh (kernel substrings: pair_heads_fwd_hand, pair_bwd_sv_kernel)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
B, N, D, classes, nh = 8, 511, 384, [2, 3, 3, 3, 3], 5
dt, dev = torch.bfloat16, "cuda"
P = N * (N + 1) // 2
ab = torch.randn(B, N, 2 * D, device=dev).to(dt)
w1 = [torch.randn(D, D, device=dev) / math.sqrt(D) for _ in classes]
w2 = [torch.randn(c, D, device=dev) / math.sqrt(D) for c in classes]
b1, b2 = torch.zeros(nh * D, device=dev), torch.zeros(14, device=dev)
wp = ops.pair_heads_pack(dt, w1, w2); wp2 = ops.pair_bwd_pack(w1); rows = ops.pair_bwd_rows(N)
tags = [torch.zeros(B, P, dtype=torch.int64, device=dev) for _ in classes]; cw = [torch.ones(c, device=dev) for c in classes]
dz = torch.empty((B * rows, nh * D), device=dev, dtype=dt)
d_ab = torch.zeros(B, N, 2 * D, device=dev); ws = ops.pair_dz_workspace(nh, D, dev, slots=256)
which = os.environ.get("WHICH", "both")
for _ in range(3):
    _, _, dl, (act, xr) = ops.pair_heads_fwd(ab, wp, b1, b2, classes, tags=tags, class_weights=cw, want_dlogits=True, want_logits=False,
                                             drop_p=0.1, drop_seed=1234, save=True)
    if which != "fwd":
        args = ops.pair_dz_args(D, classes, dl, w2, torch.ones(nh, device=dev), drop_p=0.1, drop_seed=1234)
        ops.pair_bwd_saved(ab, wp2, args, act, dz, d_ab, ws)
torch.cuda.synchronize()
print("algorithmic bytes per launch: forward writes z", B * P * nh * D * 2 / 1e6, "+ x", B * P * D * 2 / 1e6, "+ dlogits", B * P * 14 * 4 / 1e6, "MB, reads the int64 label maps",
      B * P * 5 * 8 / 1e6, "MB; backward reads z", B * P * nh * D * 2 / 1e6, "+ dlogits", B * P * 14 * 4 / 1e6, "MB and writes dz", B * P * nh * D * 2 / 1e6,
      "MB (block order:", B * rows * nh * D * 2 / 1e6, "MB) + partial rows", B * (rows // 128) * 24 * D * 4 / 1e6, "MB")
