"""The four weight-gradient GEMMs of one encoder layer: four split-k launches (+ reductions) against one grouped launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
T, H, I = 5672, 768, 3072
dt = torch.bfloat16
mk = lambda r, c: torch.randn(r, c, device="cuda").to(dt)
pairs = [(mk(T, 3 * H), mk(T, H)), (mk(T, H), mk(T, H)), (mk(T, I), mk(T, H)), (mk(T, H), mk(T, I))]
outs = [torch.empty(a.shape[1], b.shape[1], device="cuda") for a, b in pairs]
def bench(name, fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    fl = sum(2.0 * T * a.shape[1] * b.shape[1] for a, b in pairs)
    print(f"{name:40s} {us:8.1f} us   {fl / us / 1e6:7.1f} TF/s")
def single():
    for (a, b), o in zip(pairs, outs):
        ops.gemm(a, b, a_kmajor=False, b_kmajor=False, out=o)
bench("four launches (split-k + reduce)", single)
bench("one grouped launch", lambda: ops.gemm_group([(a, b, o) for (a, b), o in zip(pairs, outs)], a_kmajor=False, b_kmajor=False))
ref = [o.clone() for o in outs]
single()
print("max rel diff", max(float((o - r).abs().max() / r.abs().max()) for o, r in zip(outs, ref)))
