"""Which allocations of a steady-state train step make torch's caching allocator go to the driver (segment_alloc events, with the Python frames
that asked): python tools/dbg_segment_alloc.py [lilt]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
backbone = sys.argv[1] if len(sys.argv) > 1 else "layoutlmv3"
model, pcfg = bench.build_model("base", torch.bfloat16, backbone, 0)
model = model.cuda().set_compute_dtype(torch.bfloat16).train()
from peneo_amd.data import synthetic_rfund_batch
batches = [{k: v.cuda() for k, v in synthetic_rfund_batch(8, 512, 128, 50265, seed=s).items()} for s in range(4)]
def step(i):
    for p in model.parameters(): p.grad = None
    out = model(**batches[i % 4]); out["loss"].backward(); return out["loss"]
for i in range(3): loss = step(i)
torch.cuda.synchronize()
torch.cuda.memory._record_memory_history(max_entries=200000, context="alloc", stacks="python")
n0 = torch.cuda.memory_stats()["num_device_alloc"]
for i in range(3, 9): loss = step(i)
torch.cuda.synchronize()
print("device allocations in 6 steady-state steps:", torch.cuda.memory_stats()["num_device_alloc"] - n0)
snap = torch.cuda.memory._snapshot()
cnt = collections.Counter()
for tr in snap["device_traces"]:
    for e in tr:
        if e["action"] == "segment_alloc":
            fr = [f for f in e.get("frames", []) if "peneo_amd" in f["filename"] or "bench" in f["filename"]][:3]
            cnt[(e["size"], tuple(f"{os.path.basename(f['filename'])}:{f['line']} {f['name']}" for f in fr))] += 1
for (size, fr), n in cnt.most_common(20):
    print(f"{n:4d} x {size / 1e6:9.2f} MB  {' <- '.join(fr)}")
