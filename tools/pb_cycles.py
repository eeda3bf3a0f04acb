"""Where the waves of the wave-specialised pair_bwd kernel spend their cycles (s_memtime): loop total, s_waitcnt at the top of
the iterations (LDS-DMA landing, stores), barrier wait — producers (waves 0-3) against consumers (4-7)."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops, hip
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
dz = torch.empty(B * rows, nh * D, device=dev, dtype=dt); x = torch.empty(B * rows, D, device=dev, dtype=dt)
d_ab = torch.zeros(B, N, 2 * D, device=dev); ws = ops.pair_dz_workspace(nh, D, dev, slots=256)
args = ops.pair_dz_args(D, classes, dl, w2, scale)
ops.pair_bwd_fused(ab, wp2, b1, args, dz, x, d_ab, ws); torch.cuda.synchronize()
dbg = torch.zeros(256 * 8 * 4, dtype=torch.int64, device=dev)
lib = hip.lib(); lib.peneo_pair_bwd_debug_buffer.argtypes = [C.c_void_p]; lib.peneo_pair_bwd_debug_buffer(dbg.data_ptr())
ops.pair_bwd_fused(ab, wp2, b1, args, dz, x, d_ab, ws); torch.cuda.synchronize()
lib.peneo_pair_bwd_debug_buffer(None)
d = dbg.view(256, 8, 4).double().cpu()
for name, sl in (("producers", slice(0, 4)), ("consumers", slice(4, 8))):
    t = d[:, sl]
    print(f"{name}: loop {t[..., 0].mean():9.0f} ticks (62 iterations -> {t[..., 0].mean() / 62:6.0f} / iteration)   s_waitcnt at top {t[..., 1].mean():9.0f} ({100 * t[..., 1].mean() / t[..., 0].mean():4.1f} %)   barrier wait {t[..., 2].mean():9.0f} ({100 * t[..., 2].mean() / t[..., 0].mean():4.1f} %)")
