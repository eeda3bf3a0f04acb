"""Per-workgroup timeline of one stream-k launch (gemm_sk.hip's SK_STAMP stations, s_memrealtime at 100 MHz)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops, hip
hip.load_library()
lib = ctypes.CDLL(hip.LIB_PATH)
lib.peneo_gemm_sk_set_prof.argtypes = [ctypes.c_void_p]
names = ["start", "primed", "unit0", "pub>", "pub<", "wait>", "flag", "acq<", "ep>", "ep<", "fin ep>", "fin ep<", "end"]
mode = int(os.environ.get("MODE", 7256))
for name, m, n, k, bk in [("qkv", 5672, 2304, 768, True), ("out", 5672, 768, 768, True), ("ffn2", 5672, 768, 3072, True), ("4096^3", 4096, 4096, 4096, True)]:
    a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(n, device="cuda")
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    lib.peneo_gemm_set_sk_mode(mode)
    for _ in range(5): ops.gemm(a, w, bias=bias, out=out, split_k=1)
    prof = torch.zeros(1024, 16, dtype=torch.int64, device="cuda")
    lib.peneo_gemm_sk_set_prof(ctypes.c_void_p(prof.data_ptr()))
    ops.gemm(a, w, bias=bias, out=out, split_k=1)
    torch.cuda.synchronize()
    lib.peneo_gemm_sk_set_prof(None)
    p = prof.cpu().numpy().astype("float64")
    live = p[:, 0] > 0
    t0 = p[live, 0].min()
    print(f"== {name} [{m},{k}]x[{n}] sk{mode}: {int(live.sum())} workgroups; us after the first workgroup's start (mean / min / max over the workgroups that passed the station)")
    for i, nm in enumerate(names):
        col = p[live, i]
        ok = col > 0
        if ok.any():
            x = (col[ok] - t0) / 100.0
            print(f"   {nm:8s} n={int(ok.sum()):4d}  mean {x.mean():7.2f}  min {x.min():7.2f}  max {x.max():7.2f}")
    # per-workgroup intervals
    def iv(a_, b_):
        ok = (p[live, a_] > 0) & (p[live, b_] > 0)
        d = (p[live, b_][ok] - p[live, a_][ok]) / 100.0
        return f"{d.mean():6.2f} (max {d.max():6.2f}, n={int(ok.sum())})" if ok.any() else "-"
    print(f"   intervals: start->unit0 {iv(0, 2)} | publish {iv(3, 4)} | flag wait {iv(5, 6)} | acquire {iv(6, 7)} | acq->fin epilogue start (slab add) {iv(7, 10)} | fin epilogue {iv(10, 11)} | whole-tile epilogue {iv(8, 9)} | last station->drained {iv(11, 12)}")
