"""Big-tile GEMM (gemm_big.hip) against the 128 x 128 kernel and torch (hipBLASLt) on the encoder shapes: correctness + time."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peneo_amd import ops, hip
lib = ctypes.CDLL(hip.LIB_PATH)
DEV = "cuda"
M = int(os.environ.get("M", 5672))
shapes = [("qkv fwd", M, 2304, 768, True), ("out fwd", M, 768, 768, True), ("ffn1 fwd", M, 3072, 768, True), ("ffn2 fwd", M, 768, 3072, True),
          ("d_zi dgrad", M, 3072, 768, False), ("d_a dgrad", M, 768, 3072, False), ("d_att dgrad", M, 768, 768, False),
          ("d_x dgrad", M, 768, 2304, False), ("4096^3 NT", 4096, 4096, 4096, True), ("4096^3 NN", 4096, 4096, 4096, False),
          ("large qkv", 2442, 3072, 1024, True), ("large ffn1", 2442, 4096, 1024, True), ("large ffn2", 2442, 1024, 4096, True)]
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(0)
for name, m, n, k, bk in shapes:
    a = torch.randn(m, k, generator=g).to(DEV).to(torch.bfloat16)
    b = (torch.randn(n, k, generator=g) if bk else torch.randn(k, n, generator=g)).to(DEV).to(torch.bfloat16)
    bias = torch.randn(n, generator=g).to(DEV)
    res = torch.randn(m, n, generator=g).to(DEV).to(torch.bfloat16)
    ref = (a.float() @ (b.float().t() if bk else b.float())) + bias + res.float()
    flops = 2.0 * m * n * k
    row = f"{name:12s} [{m},{k}]x[{n}]"
    for mode in (0, 256, 384, 128, 1):
        lib.peneo_gemm_set_big_mode(mode)
        out = ops.gemm(a, b, b_kmajor=bk, bias=bias, residual=res)
        err = float((out.float() - ref).abs().max() / ref.abs().max())
        t = timeit(lambda: ops.gemm(a, b, b_kmajor=bk, bias=bias, residual=res))
        row += f" | {mode}: {t:6.1f}us {flops / t / 1e6:6.0f}TF" + ("" if err < 2e-2 else f" ERR {err:.3f}")
    bt = b.t() if bk else b
    t = timeit(lambda: torch.addmm(res, a, bt))
    row += f" | torch {t:6.1f}us {flops / t / 1e6:6.0f}TF"
    print(row, flush=True)
