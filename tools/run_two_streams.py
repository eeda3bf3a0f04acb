"""Experiment: does the step gain from two half-batches in flight on two streams (kernel tails of one overlap the other)?
Compares one 8-document step on one stream with two 4-document fwd+bwd passes issued on two streams (same weights; the
second pass accumulates into .grad, which a real implementation would avoid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from peneo_amd.data import synthetic_rfund_batch

dev = torch.device("cuda:0")
torch.manual_seed(1234)
model, pcfg = bench.build_model("base", torch.bfloat16, "layoutlmv3")
model = model.to(dev).set_compute_dtype(torch.bfloat16).train()
model.backbone.check_inputs = False
vocab = pcfg["backbone_config"]["vocab_size"]
mk = lambda B, s: {k: v.to(dev) for k, v in synthetic_rfund_batch(B, 512, 128, vocab, seed=s).items()}
b8 = [mk(8, s) for s in range(2)]
b4 = [mk(4, s) for s in range(4)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def one(i):
    for p in model.parameters():
        p.grad = None
    model(**b8[i % 2])["loss"].backward()


def two(i):
    for p in model.parameters():
        p.grad = None
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        l1 = model(**b4[(2 * i) % 4])["loss"]
    with torch.cuda.stream(s2):
        l2 = model(**b4[(2 * i + 1) % 4])["loss"]
    with torch.cuda.stream(s1):
        l1.backward()
    with torch.cuda.stream(s2):
        l2.backward()
    cur.wait_stream(s1); cur.wait_stream(s2)


def seq4(i):
    for p in model.parameters():
        p.grad = None
    model(**b4[(2 * i) % 4])["loss"].backward()
    model(**b4[(2 * i + 1) % 4])["loss"].backward()


for name, fn in (("one stream, 8 docs", one), ("two streams, 4 + 4 docs", two), ("one stream, 4 then 4 docs", seq4), ("one stream, 8 docs", one)):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 15
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / n
    print(f"{name:28s} {ms:8.3f} ms / 8 docs   {8e3 / ms:7.1f} docs/s", flush=True)
