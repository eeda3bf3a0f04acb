import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
dt = torch.bfloat16
m, n, k = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 4096, 4096))]
a = torch.randn(m, k, device="cuda").to(dt)
b = torch.randn(n, k, device="cuda").to(dt)
out = torch.empty(m, n, device="cuda", dtype=dt)
for _ in range(3):
    ops.gemm(a, b, out=out)
torch.cuda.synchronize()
