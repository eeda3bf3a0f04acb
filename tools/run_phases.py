"""Wall time of the phases of one train step (B = 8, bf16): forward, decoder backward, encoder backward."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
from seeded import layoutlmv3_config, peneo_config
from peneo_amd.model import PEneoConfig, PEneoModel
from peneo_amd.data import synthetic_rfund_batch
pcfg = peneo_config("layoutlmv3-base", layoutlmv3_config("base"))
m = PEneoModel(PEneoConfig(**{k: v for k, v in pcfg.items() if k != "model_type"})).cuda().set_compute_dtype(torch.bfloat16).train()
m.backbone.check_inputs = False
b = {k: v.cuda() for k, v in synthetic_rfund_batch(8, 512, 128, pcfg["backbone_config"]["vocab_size"], seed=1).items()}
grabbed = {}
h = m.peneo_decoder.register_forward_pre_hook(lambda mod, args, kwargs: grabbed.__setitem__("seq", kwargs["sequence_output"]), with_kwargs=True)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
res = {"fwd": [], "dec_bwd": [], "enc_bwd": []}
for it in range(6):
    for p in m.parameters(): p.grad = None
    t0 = sync()
    out = m(**b)
    t1 = sync()
    seq = grabbed["seq"]
    dec_params = [p for n, p in m.named_parameters() if n.startswith("peneo_decoder") and p.requires_grad]
    gs = torch.autograd.grad(out["loss"], [seq] + dec_params, retain_graph=True)
    t2 = sync()
    seq.backward(gs[0])
    t3 = sync()
    if it >= 2:
        res["fwd"].append(t1 - t0); res["dec_bwd"].append(t2 - t1); res["enc_bwd"].append(t3 - t2)
for k, v in res.items():
    print(f"{k:8s} {1e3 * sum(v) / len(v):7.2f} ms")
