"""Where the waves of the fused pair-space backward spend their cycles (s_memtime).
Wave-specialised kernel (D = 384): loop total, s_waitcnt at the top of the iterations (LDS-DMA landing, stores), barrier wait -
producers (waves 0-3) against consumers (4-7).
One-wave kernel (D=512 N=1023 B=2): shader-clock ticks per phase; needs the -DPB_PROF build:
    bash tools/prof_build.sh pairbwd;  gpurun -- 'PENEO_HIP_LIB=$PWD/peneo_amd/lib/libpeneo_pbprof.so D=512 N=1023 B=2 python tools/pb_cycles.py'"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops, hip
B, N, D = int(os.environ.get("B", 8)), int(os.environ.get("N", 511)), int(os.environ.get("D", 384))
classes, nh = [2, 3, 3, 3, 3], 5
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
args = ops.pair_dz_args(D, classes, dl, w2, scale, drop_p=float(os.environ.get("DROP", 0)), drop_seed=7)
ops.pair_bwd_fused(ab, wp2, b1, args, dz, x, d_ab, ws); torch.cuda.synchronize()
dbg = torch.zeros(256 * 8 * 4, dtype=torch.int64, device=dev)
lib = hip.lib(); lib.peneo_pair_bwd_debug_buffer.argtypes = [C.c_void_p]; lib.peneo_pair_bwd_debug_buffer(dbg.data_ptr())
ops.pair_bwd_fused(ab, wp2, b1, args, dz, x, d_ab, ws); torch.cuda.synchronize()
lib.peneo_pair_bwd_debug_buffer(None)
nslab = nh * D // 32
if D == 512:
    d = dbg.view(256, 4, 8).double().cpu()
    tot = d[..., 0].mean()
    print(f"one-wave kernel, D = {D}: {tot:9.0f} ticks per wave in the slab loop ({nslab} slabs -> {tot / nslab:6.0f} / slab)")
    for k, name in enumerate(("wait + barrier at the top", "weight stream issue, flush, dlogits staging", "Z (32 MFMAs at D = 512)", "E | U (VALU beside 32 MFMAs)", "dW2 sums")):
        v = d[..., 1 + k].mean()
        print(f"  {100 * v / tot:5.1f} %  {v / nslab:7.1f} ticks / slab   {name}")
else:
    d = dbg.view(256, 8, 4).double().cpu()
    for name, sl in (("producers", slice(0, 4)), ("consumers", slice(4, 8))):
        t = d[:, sl]
        print(f"{name}: loop {t[..., 0].mean():9.0f} ticks ({nslab + 2} iterations -> {t[..., 0].mean() / (nslab + 2):6.0f} / iteration)   s_waitcnt at top {t[..., 1].mean():9.0f} ({100 * t[..., 1].mean() / t[..., 0].mean():4.1f} %)   barrier wait {t[..., 2].mean():9.0f} ({100 * t[..., 2].mean() / t[..., 0].mean():4.1f} %)   fragment-read waits (PB_PROF build) {t[..., 3].mean():9.0f} ({100 * t[..., 3].mean() / t[..., 0].mean():4.1f} %)")
