"""Host enqueue time against device time of the LiLT-base eval forward and train step (is config 5 launch-bound?)."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from seeded import lilt_config, layoutlmv3_config, peneo_config
from peneo_amd.model import PEneoConfig, PEneoModel
from peneo_amd.data import synthetic_rfund_batch
which = sys.argv[1] if len(sys.argv) > 1 else "lilt"
pcfg = peneo_config("lilt-roberta-en-base", lilt_config("base")) if which == "lilt" else peneo_config("layoutlmv3-base", layoutlmv3_config("base"))
m = PEneoModel(PEneoConfig(**{k: v for k, v in pcfg.items() if k != "model_type"})).cuda().set_compute_dtype(torch.bfloat16)
m.backbone.check_inputs = False
b = synthetic_rfund_batch(8, 512, 128, pcfg["backbone_config"]["vocab_size"], seed=1)
if which == "lilt": b.pop("image", None)
b = {k: v.cuda() for k, v in b.items()}
def fwd(n=20):
    m.eval()
    with torch.no_grad():
        for _ in range(3): m(**b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): m(**b)
        th = time.perf_counter() - t0; torch.cuda.synchronize(); td = time.perf_counter() - t0
    return th / n * 1e3, td / n * 1e3
def step(n=10):
    m.train()
    for _ in range(3):
        for p in m.parameters(): p.grad = None
        m(**b)["loss"].backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        for p in m.parameters(): p.grad = None
        m(**b)["loss"].backward()
    th = time.perf_counter() - t0; torch.cuda.synchronize(); td = time.perf_counter() - t0
    return th / n * 1e3, td / n * 1e3
for _ in range(2):
    h, d = fwd(); print(f"{which} eval forward: host {h:.2f} ms, device-complete {d:.2f} ms")
    h, d = step(); print(f"{which} train step:   host {h:.2f} ms, device-complete {d:.2f} ms")
