"""Bitwise repeatability of the pair-heads forward (logits, dlogits, loss partials) and of the round-1 fused dz kernel over
several launches on the same inputs — the packed-fp32 finding of DESIGN.md §12 showed up as launch-to-launch differences."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
B, N, D, classes, nh = 8, 511, 384, [2, 3, 3, 3, 3], 5
dt, dev = torch.bfloat16, "cuda"
g = torch.Generator().manual_seed(3)
ab = torch.randn(B, N, 2 * D, generator=g).to(dev).to(dt)
P = N * (N + 1) // 2
w1 = [(torch.randn(D, D, generator=g) / math.sqrt(D)).to(dev) for _ in classes]
w2 = [(torch.randn(c, D, generator=g) / math.sqrt(D)).to(dev) for c in classes]
b1, b2 = (0.1 * torch.randn(nh * D, generator=g)).to(dev), torch.zeros(14, device=dev)
wp = ops.pair_heads_pack(dt, w1, w2)
tags = [torch.randint(0, c, (B, P), generator=g).to(dev) for c in classes]
cw = [torch.tensor([1.0, 10.0, 10.0][:c], device=dev) for c in classes]
ref = None
for rep in range(6):
    logits, partials, dlog = ops.pair_heads_fwd(ab, wp, b1, b2, classes, tags=tags, class_weights=cw, want_dlogits=True)
    torch.cuda.synchronize()
    cur = [l.clone() for l in logits] + [d.clone() for d in dlog]
    if ref is None:
        ref = cur
    else:
        print("pair_heads_fwd rep", rep, "differing logits elements per head:", [int((a != b).sum()) for a, b in zip(cur[:5], ref[:5])],
              "max |diff|", max(float((a - b).abs().max()) for a, b in zip(cur[:5], ref[:5])))
# round-1 fused dz kernel
dl = [torch.randn(P, c, generator=g).to(dev) for c in classes]
scale = torch.ones(nh, device=dev)
args = ops.pair_dz_args(D, classes, dl, w2, scale)
ref = None
for rep in range(6):
    z = torch.empty(P, nh * D, device=dev, dtype=dt)
    ops.pair_dz_fused(ab[0], 0, N, wp, b1, args, z, ops.pair_dz_workspace(nh, D, dev, slots=256))
    torch.cuda.synchronize()
    if ref is None:
        ref = z.clone()
    else:
        print("pair_dz_fused rep", rep, "differing elements:", int((z != ref).sum()), "max |diff|", float((z.float() - ref.float()).abs().max()))
