import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
B, N, D = 8, 511, 384
dt = torch.bfloat16
classes = [2, 3, 3, 3, 3]
ab = torch.randn(B, N, 2 * D, device="cuda").to(dt)
w1 = [torch.randn(D, D, device="cuda") / math.sqrt(D) for _ in classes]
w2 = [torch.randn(c, D, device="cuda") / math.sqrt(D) for c in classes]
b1, b2 = torch.zeros(5 * D, device="cuda"), torch.zeros(14, device="cuda")
wp = ops.pair_heads_pack(dt, w1, w2)
nwg = B * 511
dbg = torch.zeros(nwg, 4, dtype=torch.int64, device="cuda")
ops.pair_heads_fwd(ab, wp, b1, b2, classes)
torch.cuda.synchronize()
os.environ["PENEO_PAIR_DBG_PTR"] = str(dbg.data_ptr())
ops.pair_heads_fwd(ab, wp, b1, b2, classes)
torch.cuda.synchronize()
d = dbg.cpu().double()
pro, loop, epi = d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2]
print(f"s_memtime ticks per WG (median): prologue {pro.median():.0f}  loop {loop.median():.0f}  epilogue {epi.median():.0f}  total {(d[:,3]-d[:,0]).median():.0f}")
print(f"kernel span (first start -> last end): {(d[:,3].max() - d[:,0].min()):.0f} ticks; per-slab loop ticks {loop.median()/60:.0f}")
