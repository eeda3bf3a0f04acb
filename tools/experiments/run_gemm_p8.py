"""gemm_p8.hip (256 x 256, staggered wave groups) against the round-3 kernels and torch (hipBLASLt) on the encoder shapes:
correctness (asymmetric random operands, fp32 reference, bit-repeatability over 10 launches = race screen) + time.
    python tools/run_gemm_p8.py [quick]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peneo_amd import ops, hip
lib = ctypes.CDLL(hip.LIB_PATH)
DEV = "cuda"
M = int(os.environ.get("M", 5672))
shapes = [("qkv fwd", M, 2304, 768, True, (1,)), ("out fwd", M, 768, 768, True, (1, 2, 3)), ("ffn1 fwd", M, 3072, 768, True, (1,)),
          ("ffn2 fwd", M, 768, 3072, True, (1, 2, 3, 4)),
          ("d_zi dgrad", M, 3072, 768, False, (1,)), ("d_a dgrad", M, 768, 3072, False, (1, 3)), ("d_att dgrad", M, 768, 768, False, (1, 3)),
          ("d_x dgrad", M, 768, 2304, False, (1, 3)), ("4096^3 NT", 4096, 4096, 4096, True, (1,)), ("4096^3 NN", 4096, 4096, 4096, False, (1,)),
          ("8192^3 NT", 8192, 8192, 8192, True, (1,)),
          ("ragged", 1000, 520, 192, True, (1, 3)), ("ragged NN", 1000, 520, 192, False, (1,)),
          ("large qkv", 2442, 3072, 1024, True, (1,)), ("large ffn2", 2442, 1024, 4096, True, (1, 2, 4))]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    shapes = [s for s in shapes if s[0] in ("qkv fwd", "ffn2 fwd", "d_zi dgrad", "4096^3 NT", "4096^3 NN", "ragged", "ragged NN")]
def timeit(fn, n=20, reps=5):
    """us per call from a captured graph of n calls (the Python launch loop is host-bound below ~40 us per kernel)"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(n): fn()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps): gr.replay()
        e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3
g = torch.Generator().manual_seed(0)
for name, m, n, k, bk, splits in shapes:
    a = torch.randn(m, k, generator=g).to(DEV).to(torch.bfloat16)
    b = (torch.randn(n, k, generator=g) if bk else torch.randn(k, n, generator=g)).to(DEV).to(torch.bfloat16)
    bias = torch.randn(n, generator=g).to(DEV)
    res = torch.randn(m, n, generator=g).to(DEV).to(torch.bfloat16)
    ref = (a.float() @ (b.float().t() if bk else b.float())) + bias + res.float()
    flops = 2.0 * m * n * k
    row = f"{name:12s} [{m},{k}]x[{n}]"
    lib.peneo_gemm_set_p8_mode(0)
    t = timeit(lambda: ops.gemm(a, b, b_kmajor=bk, bias=bias, residual=res))
    row += f" | r03 {t:6.1f}us {flops / t / 1e6:5.0f}TF"
    lib.peneo_gemm_set_p8_mode(2)
    for flags in (0, 1):
      lib.peneo_gemm_set_p8_flags(flags)
      for sk in splits:
        out = ops.gemm(a, b, b_kmajor=bk, bias=bias, residual=res, split_k=sk)
        err = float((out.float() - ref).abs().max() / ref.abs().max())
        same = all(torch.equal(out, ops.gemm(a, b, b_kmajor=bk, bias=bias, residual=res, split_k=sk)) for _ in range(10))
        t = timeit(lambda: ops.gemm(a, b, b_kmajor=bk, bias=bias, residual=res, split_k=sk))
        row += f" | p8{'r' if flags else ''}/k{sk} {t:6.1f}us {flops / t / 1e6:5.0f}TF" + ("" if err < 2e-2 else f" ERR {err:.3f}") + ("" if same else " NONDET")
    lib.peneo_gemm_set_p8_mode(0)
    bt = b.t() if bk else b
    t = timeit(lambda: torch.addmm(res, a, bt))
    row += f" | torch {t:6.1f}us {flops / t / 1e6:5.0f}TF"
    print(row, flush=True)
