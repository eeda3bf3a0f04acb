"""CPU oracle, one base-size document fwd + bwd on 32 host threads: the reference's as-executed form (materialised [N, N, 2D]
handshaking input, one-hot bias GEMMs) against the algebraically reduced form — the two figures of bench.py's cpu_baseline."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
from oracle import peneo_oracle as O
from peneo_amd.data import synthetic_rfund_batch
from peneo_amd.model import PEneoConfig, PEneoModel
torch.set_num_threads(32)
model, pcfg = None, None
from tests.golden.seeded import layoutlmv3_config, peneo_config
pcfg = peneo_config("layoutlmv3-base", layoutlmv3_config("base"))
m = PEneoModel(PEneoConfig(**{k: v for k, v in pcfg.items() if k != "model_type"}))
sd = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() and not k.endswith("_loss.weight") else v) for k, v in m.state_dict().items()}
batch = synthetic_rfund_batch(1, 512, 128, 50265, seed=7)
for ax in (True, True, False):
    for v in sd.values():
        if v.is_floating_point() and v.requires_grad: v.grad = None
    t0 = time.perf_counter()
    out = O.peneo_forward(sd, pcfg, batch, training=False, as_executed=ax)
    t1 = time.perf_counter()
    out["loss"].backward()
    t2 = time.perf_counter()
    print("as_executed", ax, f"fwd {t1 - t0:.2f}s  bwd {t2 - t1:.2f}s  total {t2 - t0:.2f}s", flush=True)
