import math, sys, os
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
train = os.environ.get("TRAIN", "1") == "1"    # the launch bench.py times in a train step (round 3): class-weighted CE + dlogits with
                                               # the classifier dropout on and NO logit maps written; TRAIN=0: eval, logits only
P = N * (N + 1) // 2
tags = [torch.zeros(B, P, dtype=torch.int64, device="cuda") for _ in classes] if train else None
cw = [torch.ones(c, device="cuda") for c in classes] if train else None
for _ in range(3):
    ops.pair_heads_fwd(ab, wp, b1, b2, classes, tags=tags, class_weights=cw, want_dlogits=train, want_logits=not train,
                       drop_p=0.1 if train else 0.0, drop_seed=1234)
torch.cuda.synchronize()
