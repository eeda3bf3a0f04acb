"""Eval forward only (B = 8, bf16): the 'encoder + pair-head forward' target of BASELINE.md.  For rocprofv3 --kernel-trace."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from seeded import layoutlmv3_config, lilt_config, peneo_config
from peneo_amd.model import PEneoConfig, PEneoModel
from peneo_amd.data import synthetic_rfund_batch
LILT = os.environ.get("BACKBONE", "layoutlmv3") == "lilt"
SIZE = os.environ.get("SIZE", "base"); SEQ = int(os.environ.get("SEQ", 512)); LINES = int(os.environ.get("LINES", 128)); DOCS = int(os.environ.get("DOCS", 8))
pcfg = peneo_config("lilt-roberta-en-base", lilt_config("base")) if LILT else peneo_config("layoutlmv3-base", layoutlmv3_config(SIZE))
m = PEneoModel(PEneoConfig(**{k: v for k, v in pcfg.items() if k != "model_type"})).cuda().set_compute_dtype(torch.bfloat16).eval()
m.backbone.check_inputs = False
bs = [{k: v.cuda() for k, v in synthetic_rfund_batch(DOCS, SEQ, LINES, 50265, seed=s).items() if not (LILT and k == "image")} for s in range(3)]
with torch.no_grad():
    for i in range(3): m(**bs[i])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = int(os.environ.get("N", "10"))
    for i in range(n): m(**bs[i % 3])
    torch.cuda.synchronize()
print(f"eval forward {(time.perf_counter() - t0) / n * 1e3:.3f} ms per {DOCS} documents")
