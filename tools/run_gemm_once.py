"""One encoder GEMM shape, a few launches (for rocprofv3 --pmc): SHAPE = qkv | out | ffn1 | ffn2 (forward, M = 5672) or
d_zi | d_a | d_x (dgrad); PENEO_GEMM_BIG=0 selects the 128 x 128 kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peneo_amd import ops
shape = os.environ.get("SHAPE", "ffn1")
M = 5672
N, K, bk = {"qkv": (2304, 768, True), "out": (768, 768, True), "ffn1": (3072, 768, True), "ffn2": (768, 3072, True),
            "d_zi": (3072, 768, False), "d_a": (768, 3072, False), "d_x": (768, 2304, False)}[shape]
g = torch.Generator().manual_seed(0)
a = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16)
b = (torch.randn(N, K, generator=g) if bk else torch.randn(K, N, generator=g)).cuda().to(torch.bfloat16)
bias = torch.randn(N, generator=g).cuda()
for _ in range(6):
    ops.gemm(a, b, b_kmajor=bk, bias=bias)
torch.cuda.synchronize()
