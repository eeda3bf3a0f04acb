import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
B, N, D = int(os.environ.get("B", 8)), int(os.environ.get("N", 511)), int(os.environ.get("D", 384))
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
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.pair_heads_fwd(ab, wp, b1, b2, classes, tags=tags, class_weights=cw, want_dlogits=train, want_logits=not train,
                       drop_p=0.1 if train else 0.0, drop_seed=1234)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"pair_heads_fwd {'train' if train else 'eval'} B={B} N={N} D={D}: {ms * 1e3:8.1f} us   {2.0 * B * P * 5 * D * (D + 3) / ms / 1e9:7.1f} TF/s")
