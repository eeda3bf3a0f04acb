"""The train step on the default stream against a HIGH-priority stream (side streams stay at normal priority)."""
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
def run(stream, n=20):
    with torch.cuda.stream(stream):
        for i in range(n + 3):
            if i == 3: torch.cuda.synchronize(); t0 = time.perf_counter()
            for p in m.parameters(): p.grad = None
            m(**b)["loss"].backward()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
hi = torch.cuda.Stream(priority=-1)
lo = torch.cuda.Stream(priority=0)
for _ in range(3):
    print(f"default stream {run(torch.cuda.default_stream()):.3f} ms   normal-priority stream {run(lo):.3f} ms   high-priority stream {run(hi):.3f} ms")
