"""Encoder / decoder GEMM shapes under the persistent stream-k launch (gemm_sk.hip) against the tiled kernels and torch.mm
(the vendor library): us per launch (HIP events, 30 launches after 5 warm-up), TFLOP/s.  Modes: sk 0 = tiled kernels (r05),
128 / 256 = stream-k tile forced."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops, hip
hip.load_library()
lib = ctypes.CDLL(hip.LIB_PATH)

def bench(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

M = int(os.environ.get("M", 5672))
shapes = [("qkv fwd", M, 2304, 768, True), ("out fwd", M, 768, 768, True), ("ffn1 fwd", M, 3072, 768, True), ("ffn2 fwd", M, 768, 3072, True),
          ("dec z", 130816, 1920, 384, True),
          ("large qkv", 2442, 3072, 1024, True), ("large ffn1", 2442, 4096, 1024, True), ("large ffn2", 2442, 1024, 4096, True),
          ("4096^3", 4096, 4096, 4096, True)]
modes = [int(x) for x in os.environ.get("MODES", "0,1,5256,4256,5128,105128").split(",")]
for name, m, n, k, bk in shapes:
    a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") if bk else torch.randn(k, n, device="cuda")).to(torch.bfloat16) * 0.05
    bias = torch.randn(n, device="cuda")
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * m * n * k
    line = f"{name:12s} [{m},{k}]x[{n}]"
    ref = None
    for md in modes:
        lib.peneo_gemm_set_sk_mode(md)
        t = bench(lambda: ops.gemm(a, w, b_kmajor=bk, bias=bias, out=out, split_k=1))
        if ref is None: ref = out.float().clone()
        err = float((out.float() - ref).abs().max() / ref.abs().max())
        line += f" | sk{md}: {t:6.1f}us {fl / t / 1e6:5.0f}TF e{err:.0e}"
    wt = w.t() if bk else w
    t = bench(lambda: torch.mm(a, wt))
    line += f" | torch {t:6.1f}us {fl / t / 1e6:5.0f}TF"
    print(line, flush=True)
lib.peneo_gemm_set_sk_mode(1)
