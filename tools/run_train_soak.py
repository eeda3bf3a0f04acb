"""Soak: N optimizer steps of the base model (train mode, fused AdamW, four resident batches) - the loss must stay finite and fall."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from seeded import layoutlmv3_config, peneo_config
from peneo_amd.model import PEneoConfig, PEneoModel
from peneo_amd.data import synthetic_rfund_batch
from peneo_amd.optim import FusedAdamW
torch.manual_seed(0)
torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
pcfg = peneo_config("layoutlmv3-base", layoutlmv3_config("base"))
m = PEneoModel(PEneoConfig(**{k: v for k, v in pcfg.items() if k != "model_type"})).cuda().set_compute_dtype(torch.bfloat16).train()
m.backbone.check_inputs = False
bs = [{k: v.cuda() for k, v in synthetic_rfund_batch(8, 512, 128, 50265, seed=s, ragged=(s % 2 == 1)).items()} for s in range(4)]
opt = FusedAdamW(m.parameters(), lr=2e-5, weight_decay=0.01)
N = int(os.environ.get("N", "120"))
losses = []
t0 = time.perf_counter()
for i in range(N):
    opt.zero_grad(set_to_none=True)
    loss = m(**bs[i % 4])["loss"]
    loss.backward()
    opt.step()
    if i % 10 == 0 or i == N - 1:
        losses.append(float(loss))
torch.cuda.synchronize()
print("losses", [round(x, 4) for x in losses])
print(f"{N} steps in {time.perf_counter() - t0:.1f} s; finite: {all(x == x and abs(x) < 1e4 for x in losses)}; fell: {losses[-1] < losses[0]}")
print(f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
