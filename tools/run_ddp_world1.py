"""DDP over RCCL with a world of ONE rank on the real GPU: exercises the bucketed bf16 all-reduce hooks and RCCL's own
stream beside the four streams of the step (the multi-GPU bench cannot be launched from here).  Prints docs/s with and
without the wrapper."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
import torch
import torch.distributed as dist
from seeded import layoutlmv3_config, peneo_config
from peneo_amd.model import PEneoConfig, PEneoModel
from peneo_amd.data import synthetic_rfund_batch
from peneo_amd.parallel import wrap_data_parallel
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
pcfg = peneo_config("layoutlmv3-base", layoutlmv3_config("base"))
m = PEneoModel(PEneoConfig(**{k: v for k, v in pcfg.items() if k != "model_type"})).cuda().set_compute_dtype(torch.bfloat16).train()
m.backbone.check_inputs = False
b = {k: v.cuda() for k, v in synthetic_rfund_batch(8, 512, 128, pcfg["backbone_config"]["vocab_size"], seed=1).items()}
def run(net, tag, n=8):
    def step():
        for p in m.parameters(): p.grad = None
        out = net(**b); out["loss"].backward(); return out["loss"]
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): loss = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{tag:28s} {8 / dt:7.1f} docs/s  {dt * 1e3:6.2f} ms/step  loss {float(loss):.5f}")
run(m, "plain module")
net = wrap_data_parallel(m, device_ids=[0], impl="flat")
run(net, "flat all-reduce, world 1")
g_flat = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
run(m, "plain module again")
net = wrap_data_parallel(m, device_ids=[0], impl="ddp")
run(net, "torch DDP over RCCL, world 1")
dist.destroy_process_group()
