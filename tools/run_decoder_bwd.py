"""One chunk of the decoder backward at the model's shape: per-kernel timing."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
N, D = 511, 384
classes = [2, 3, 3, 3, 3]
nh = 5
dt = torch.bfloat16
dev = "cuda"
ab = torch.randn(N, 2 * D, device=dev).to(dt)
i0, i1 = 0, 69
p0, p1 = 0, i1 * N - i1 * (i1 - 1) // 2
npairs = p1 - p0
print("npairs", npairs)
x = torch.empty(npairs, D, device=dev, dtype=dt)
w1 = (torch.randn(nh * D, D, device=dev) / math.sqrt(D)).to(dt)
b1 = torch.zeros(nh * D, device=dev)
w2 = [torch.randn(c, D, device=dev) for c in classes]
dl = [torch.randn(npairs, c, device=dev) for c in classes]
scale = torch.ones(nh, device=dev)
z = torch.empty(npairs, nh * D, device=dev, dtype=dt)
dx = torch.empty(npairs, D, device=dev, dtype=dt)
dW = torch.zeros(nh * D, D, device=dev)
dab = torch.zeros(N, 2 * D, device=dev)
ws = ops.pair_dz_workspace(nh, D, dev)
dza = ops.pair_dz_args(D, classes, dl, w2, scale)
def bench(name, fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:36s} {e0.elapsed_time(e1) / n * 1e3:8.1f} us")
fl = 2.0 * npairs * nh * D * D
bench("pair_x_fwd", lambda: ops.pair_x_fwd(ab, i0, i1, x))
bench("gemm z (plain)", lambda: ops.gemm(x, w1, bias=b1, out=z))
bench("gemm z + dz epilogue", lambda: ops.gemm(x, w1, bias=b1, out=z, pair_dz=dza, pair_dz_ws=ws))
w1l = [w1[h * D:(h + 1) * D].float() for h in range(nh)]
wp = ops.pair_heads_pack(dt, w1l, w2)
bench("pair_dz_fused (no x / z)", lambda: ops.pair_dz_fused(ab, i0, i1, wp, b1, dza, z, ws))
if os.environ.get("FULL"):
    i1f = N
    npf = N * (N + 1) // 2
    zf = torch.empty(npf, nh * D, device=dev, dtype=dt)
    dlf = [torch.randn(npf, c, device=dev) for c in classes]
    dzaf = ops.pair_dz_args(D, classes, dlf, w2, scale)
    bench("pair_dz_fused whole document", lambda: ops.pair_dz_fused(ab, 0, N, wp, b1, dzaf, zf, ws))
    xf_ = torch.empty(npf, D, device=dev, dtype=dt)
    ops.pair_x_fwd(ab, 0, N, xf_)
    bench("gemm z + dz epilogue whole document", lambda: ops.gemm(xf_, w1, bias=b1, out=zf, pair_dz=dzaf, pair_dz_ws=ws))
bench("pair_dz (separate)", lambda: ops.pair_dz(z, npairs, D, classes, dl, w2, ws, scale))
bench("gemm dW += dz^T x", lambda: ops.gemm(z, x, a_kmajor=False, b_kmajor=False, out=dW, accumulate=True))
bench("gemm dx = dz W1", lambda: ops.gemm(z, w1, b_kmajor=False, out=dx))
bench("pair_x_bwd", lambda: ops.pair_x_bwd(ab, i0, i1, dx, dab))
print("GEMM flops each:", fl / 1e9, "GF")
