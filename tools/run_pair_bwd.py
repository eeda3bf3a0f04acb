"""Times peneo_pair_bwd_fused (whole batch) and the dW1 GEMM it leaves, at config 2 (B = 8, N = 511, D = 384) or, with
B=2 N=1023 D=512, config 4 (the one-wave kernel); DROP=0.1 with the classifier dropout.
Ablation builds: PENEO_HIP_LIB=<lib built with -DPB_ABLATE=n>."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
B = int(os.environ.get("B", 8))
N, D, classes, nh = int(os.environ.get("N", 511)), int(os.environ.get("D", 384)), [2, 3, 3, 3, 3], 5
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
args = ops.pair_dz_args(D, classes, dl, w2, scale, drop_p=float(os.environ.get("DROP", 0)), drop_seed=7)
dW1 = torch.zeros(nh * D, D, device=dev)

def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

t = timeit(lambda: ops.pair_bwd_fused(ab, wp2, b1, args, dz, x, d_ab, ws))
fl = 2 * 2.0 * B * rows * nh * D * D
print(f"pair_bwd_fused  B={B} N={N} D={D}: {t * 1e3:8.1f} us  = {t * 1e3 / B:7.1f} us/doc   {fl / t / 1e9:7.1f} TF/s (z + du, block rows)")
t2 = timeit(lambda: ops.gemm(dz, x, a_kmajor=False, b_kmajor=False, out=dW1))
print(f"dW1 = dz^T x    K={B * rows}: {t2 * 1e3:8.1f} us  = {t2 * 1e3 / B:7.1f} us/doc   {fl / 2 / t2 / 1e9:7.1f} TF/s")
