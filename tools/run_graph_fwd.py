"""Eval forward captured into a HIP graph (torch.cuda.CUDAGraph): how much of the forward is inter-kernel gaps?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
from seeded import layoutlmv3_config, peneo_config
from peneo_amd.model import PEneoConfig, PEneoModel
from peneo_amd.data import synthetic_rfund_batch
pcfg = peneo_config("layoutlmv3-base", layoutlmv3_config("base"))
m = PEneoModel(PEneoConfig(**{k: v for k, v in pcfg.items() if k != "model_type"})).cuda().set_compute_dtype(torch.bfloat16).eval()
m.backbone.check_inputs = False
b = {k: v.cuda() for k, v in synthetic_rfund_batch(8, 512, 128, pcfg["backbone_config"]["vocab_size"], seed=1).items()}
b2 = {k: v.cuda() for k, v in synthetic_rfund_batch(8, 512, 128, pcfg["backbone_config"]["vocab_size"], seed=2).items()}
torch.set_grad_enabled(False)
for _ in range(3): out = m(**b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): out = m(**b)
torch.cuda.synchronize()
print(f"eager eval forward: {(time.perf_counter() - t0) * 100:.3f} ms")
ref = {k: v.clone() for k, v in out.items() if isinstance(v, torch.Tensor)}
static = {k: v.clone() for k, v in b.items()}
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): m(**static)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    gout = m(**static)
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): g.replay()
torch.cuda.synchronize()
print(f"graph replay eval forward: {(time.perf_counter() - t0) * 100:.3f} ms")
err = max(float((gout[k].float() - ref[k].float()).abs().max()) for k in ref if k.endswith("outputs"))
print("max |graph - eager| on logits:", err)
for k in static: static[k].copy_(b2[k])
g.replay(); torch.cuda.synchronize()
o2 = m(**b2)
err2 = max(float((gout[k].float() - o2[k].float()).abs().max()) for k in ref if k.endswith("outputs"))
print("new inputs through the same graph, max diff vs eager:", err2)
