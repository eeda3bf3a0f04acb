"""60 real training steps (forward, backward, FusedAdamW) at config 2: loss and allocator figures every 10 steps — nothing the
held side-stream joins keep alive may accumulate."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
from peneo_amd.data import synthetic_rfund_batch
from peneo_amd.optim import FusedAdamW, peneo_param_groups
dev = torch.device("cuda:0")
torch.manual_seed(1)
model, pcfg = bench.build_model("base", torch.bfloat16, "layoutlmv3")
model = model.to(dev).set_compute_dtype(torch.bfloat16).train()
model.backbone.check_inputs = False
vocab = pcfg["backbone_config"]["vocab_size"]
bs = [{k: v.to(dev) for k, v in synthetic_rfund_batch(8, 512, 128, vocab, seed=s).items()} for s in range(3)]
opt = FusedAdamW(peneo_param_groups(model, 5e-5, 0.01, 30.0))
for i in range(60):
    out = model(**bs[i % 3])
    opt.zero_grad()
    out.loss.backward()
    opt.step()
    if i % 10 == 9:
        torch.cuda.synchronize()
        print(i, f"loss {float(out.loss.detach()):.4f}", f"alloc {torch.cuda.memory_allocated() / 2**30:.2f} GiB", f"peak {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", f"reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB", flush=True)
