"""Is the step launch-bound?  Host time to *issue* a step (no sync) against the time the GPU needs for it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
from seeded import layoutlmv3_config, peneo_config
from peneo_amd.model import PEneoConfig, PEneoModel
from peneo_amd.data import synthetic_rfund_batch
from peneo_amd import ops
pcfg = peneo_config("layoutlmv3-base", layoutlmv3_config("base"))
m = PEneoModel(PEneoConfig(**{k: v for k, v in pcfg.items() if k != "model_type"})).cuda().set_compute_dtype(torch.bfloat16).train()
m.backbone.check_inputs = False
b = {k: v.cuda() for k, v in synthetic_rfund_batch(8, 512, 128, pcfg["backbone_config"]["vocab_size"], seed=1).items()}
def step():
    for p in m.parameters(): p.grad = None
    out = m(**b); out["loss"].backward()
for _ in range(3): step()
torch.cuda.synchronize()
for name, fn in (("train step", step), ("eval forward", lambda: m(**b))):
    if name == "eval forward":
        m.eval(); torch.set_grad_enabled(False); fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    t_issue = (time.perf_counter() - t0) / 5
    torch.cuda.synchronize()
    t_total = (time.perf_counter() - t0) / 5
    print(f"{name}: host issue {t_issue * 1e3:.2f} ms, wall {t_total * 1e3:.2f} ms per step")

# how far ahead of the GPU is the host at the phase boundaries of a train step?  (lead ~ 0 => the GPU waits for launches)
m.train(); torch.set_grad_enabled(True)
for _ in range(2): step()
torch.cuda.synchronize()
marks = []
def mark(tag):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((tag, time.perf_counter(), e))
mark("start")
for i in range(4):
    for p in m.parameters(): p.grad = None
    mark(f"s{i} fwd issue begins")
    out = m(**b)
    mark(f"s{i} fwd issued")
    out["loss"].backward()
    mark(f"s{i} bwd issued")
torch.cuda.synchronize()
t0h, e0 = marks[0][1], marks[0][2]
for tag, th, e in marks[1:]:
    tg = e0.elapsed_time(e)
    print(f"{tag:22s} host {1e3 * (th - t0h):8.2f} ms   gpu {tg:8.2f} ms   gpu-behind-host {tg - 1e3 * (th - t0h):7.2f} ms")
