"""Ablations of gemm_p8 at 4096^3 / 8192^3 and QKV: flags bit1 = no LDS-DMA in the loop, bit2 = no MFMA (results are wrong)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peneo_amd import ops, hip
lib = ctypes.CDLL(hip.LIB_PATH)
def timeit(fn, n=20, reps=5):
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
lib.peneo_gemm_set_p8_mode(2)
for name, m, n, k in [("4096^3", 4096, 4096, 4096), ("4096x4096xK8192", 4096, 4096, 8192), ("qkv", 5672, 2304, 768), ("qkv K=3072", 5672, 2304, 3072)]:
    a = torch.randn(m, k, generator=g).cuda().to(torch.bfloat16)
    b = torch.randn(n, k, generator=g).cuda().to(torch.bfloat16)
    out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    row = f"{name:16s}"
    for flags, lab in [(1, "4ph"), (3, "noDMA"), (5, "noMMA"), (7, "neither"), (9, "2ph"), (11, "2ph noDMA"), (13, "2ph noMMA"), (15, "2ph neither")]:
        lib.peneo_gemm_set_p8_flags(flags)
        t = timeit(lambda: ops.gemm(a, b, out=out))
        row += f" | {lab} {t:7.1f}us ({t / (k / 64):5.2f}/kt)"
    lib.peneo_gemm_set_p8_flags(1)
    print(row, flush=True)
