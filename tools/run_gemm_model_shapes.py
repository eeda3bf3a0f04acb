"""The four forward GEMMs of an encoder layer with the epilogues the model uses (eval: no dropout, no pre-activation store)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
from peneo_amd.hip import ACT_GELU, ACT_NONE
def bench(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 5672
for name, N, K, act, res in (("QKV", 2304, 768, ACT_NONE, False), ("O", 768, 768, ACT_NONE, True), ("FFN1", 3072, 768, ACT_GELU, False), ("FFN2", 768, 3072, ACT_NONE, True)):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if res else None
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t_plain = bench(lambda: ops.gemm(a, w, out=out))
    t_bias = bench(lambda: ops.gemm(a, w, bias=bias, out=out))
    t_model = bench(lambda: ops.gemm(a, w, bias=bias, act=act, residual=r, out=out))
    t_train = bench(lambda: ops.gemm(a, w, bias=bias, act=act, residual=r, out=out, preact=pre if act != ACT_NONE else None,
                                     drop_p=0.1 if res else 0.0, drop_seed=3))
    fl = 2.0 * M * N * K
    print(f"{name:5s} plain {t_plain:5.1f}  +bias {t_bias:5.1f}  model(eval) {t_model:5.1f} us = {fl / t_model / 1e6:5.0f} TF   train {t_train:5.1f} us")
