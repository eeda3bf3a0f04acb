"""Calibration workload for the HBM PMC counters: copy2d of a 1 GiB bf16 matrix (reads 1 GiB, writes 1 GiB per launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
x = torch.randn(65536, 8192, device="cuda").to(torch.bfloat16)   # 1 GiB
y = torch.empty_like(x)
for _ in range(3):
    ops.copy2d(x, y)
torch.cuda.synchronize()
