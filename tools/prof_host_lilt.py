"""cProfile of the host side of LiLT train steps at a size where the device never limits (B = 1, S = 64): where the launch loop spends its time."""
import sys, os, time, torch, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from seeded import lilt_config, layoutlmv3_config, peneo_config
from peneo_amd.model import PEneoConfig, PEneoModel
from peneo_amd.data import synthetic_rfund_batch
which = sys.argv[1] if len(sys.argv) > 1 else "lilt"
pcfg = peneo_config("lilt-roberta-en-base", lilt_config("base")) if which == "lilt" else peneo_config("layoutlmv3-base", layoutlmv3_config("base"))
m = PEneoModel(PEneoConfig(**{k: v for k, v in pcfg.items() if k != "model_type"})).cuda().set_compute_dtype(torch.bfloat16).train()
m.backbone.check_inputs = False
b = synthetic_rfund_batch(1, 64, 16, pcfg["backbone_config"]["vocab_size"], seed=1)
if which == "lilt": b.pop("image", None)
b = {k: v.cuda() for k, v in b.items()}
def step():
    for p in m.parameters(): p.grad = None
    m(**b)["loss"].backward()
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); print(f"{which}: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per step at B = 1, S = 64 (host-bound)")
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
