"""Where the waves of the hand-interleaved pair_heads_fwd kernel spend their cycles (s_memtime): the vm wait, the barrier and the DMA issue at
the top of a slab against the slab bodies.  Needs the -DPH_PROF build:  bash tools/prof_build.sh pairfwd;
gpurun -- 'PENEO_HIP_LIB=$PWD/peneo_amd/lib/libpeneo_phprof.so python tools/ph_cycles.py'   (TRAIN=0: eval)"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops, hip
B, N, D = int(os.environ.get("B", 8)), int(os.environ.get("N", 511)), int(os.environ.get("D", 384))
dt, classes = torch.bfloat16, [2, 3, 3, 3, 3]
ab = torch.randn(B, N, 2 * D, device="cuda").to(dt)
w1 = [torch.randn(D, D, device="cuda") / math.sqrt(D) for _ in classes]
w2 = [torch.randn(c, D, device="cuda") / math.sqrt(D) for c in classes]
b1, b2 = torch.zeros(5 * D, device="cuda"), torch.zeros(14, device="cuda")
wp = ops.pair_heads_pack(dt, w1, w2)
train = os.environ.get("TRAIN", "1") == "1"
P = N * (N + 1) // 2
tags = [torch.zeros(B, P, dtype=torch.int64, device="cuda") for _ in classes] if train else None
cw = [torch.ones(c, device="cuda") for c in classes] if train else None
run = lambda: ops.pair_heads_fwd(ab, wp, b1, b2, classes, tags=tags, class_weights=cw, want_dlogits=train, want_logits=not train, drop_p=0.1 if train else 0.0, drop_seed=1)
run(); torch.cuda.synchronize()
dbg = torch.zeros(256 * 8 * 4, dtype=torch.int64, device="cuda")
lib = hip.lib(); lib.peneo_pair_fwd_prof_buffer.argtypes = [C.c_void_p]; assert lib.peneo_pair_fwd_prof_buffer(dbg.data_ptr()) == 0
run(); torch.cuda.synchronize()
d = dbg.view(256, 8, 4).double().cpu()
nslab = 5 * D // 32
tot = d.sum(-1).mean()
print(f"{'train' if train else 'eval'}: {tot:9.0f} ticks per wave over {nslab} slabs = {tot / nslab:6.0f} / slab")
for k, name in enumerate(("wait for the slab's LDS-DMA pieces", "barrier", "DMA issue of slab + 2", "slab body (MFMAs + the previous slab's epilogue)")):
    print(f"  {100 * d[..., k].mean() / tot:5.1f} %  {d[..., k].mean() / nslab:7.1f} ticks / slab   {name}")
