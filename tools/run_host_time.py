"""Host time against device time of the benchmark step: the launch loop of N steps is timed twice - until the host has queued
the last launch, and until the device has finished it.  host ~ device means the step is launch-bound."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from seeded import layoutlmv3_config, peneo_config
from peneo_amd.model import PEneoConfig, PEneoModel
from peneo_amd.data import synthetic_rfund_batch
pcfg = peneo_config("layoutlmv3-base", layoutlmv3_config("base"))
m = PEneoModel(PEneoConfig(**{k: v for k, v in pcfg.items() if k != "model_type"})).cuda().set_compute_dtype(torch.bfloat16).train()
m.backbone.check_inputs = False
b = {k: v.cuda() for k, v in synthetic_rfund_batch(8, 512, 128, 50265, seed=1).items()}
def run(n=20, fwd_only=False):
    for i in range(3):
        for p in m.parameters(): p.grad = None
        m(**b)["loss"].backward()
    torch.cuda.synchronize(); t0 = time.perf_counter(); tf = 0.0
    for i in range(n):
        for p in m.parameters(): p.grad = None
        t1 = time.perf_counter()
        out = m(**b)
        tf += time.perf_counter() - t1
        if not fwd_only:
            out["loss"].backward()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return th / n * 1e3, (time.perf_counter() - t0) / n * 1e3, tf / n * 1e3
for _ in range(3):
    h, d, f = run()
    print(f"host {h:.2f} ms/step (forward part {f:.2f}), device-complete {d:.2f} ms/step")
