"""Where the HOST time of a train step goes: wall time of every autograd stage's forward / backward body (the Python + FFI work
that enqueues its kernels), of the loss.backward() call as a whole, and of zeroing the gradients."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from seeded import layoutlmv3_config, peneo_config
from peneo_amd.model import PEneoConfig, PEneoModel
from peneo_amd.model import modeling_layoutlmv3 as ML, peneo_decoder as PD, modeling_peneo as MP
from peneo_amd.data import synthetic_rfund_batch
pcfg = peneo_config("layoutlmv3-base", layoutlmv3_config("base"))
m = PEneoModel(PEneoConfig(**{k: v for k, v in pcfg.items() if k != "model_type"})).cuda().set_compute_dtype(torch.bfloat16).train()
m.backbone.check_inputs = False
BS, SL = int(os.environ.get("B", 8)), int(os.environ.get("S", 512))
b = {k: v.cuda() for k, v in synthetic_rfund_batch(BS, SL, max(4, SL // 4), 50265, seed=1).items()}
print(f"B = {BS}, S = {SL}")
acc = {}
def wrap(cls, name):
    for which in ("forward", "backward"):
        fn = getattr(cls, which)
        def make(fn, key):
            def inner(*a, **k):
                t = time.perf_counter()
                r = fn(*a, **k)
                acc[key] = acc.get(key, 0.0) + time.perf_counter() - t
                return r
            return staticmethod(inner)
        setattr(cls, which, make(fn, f"{name}.{which}"))
stages = {"embed": ML._EmbedStage, "layer": ML._LayerStage}
for mod in (PD, MP):
    for n in dir(mod):
        o = getattr(mod, n)
        if isinstance(o, type) and issubclass(o, torch.autograd.Function) and o is not torch.autograd.Function:
            stages[n] = o
for n, c in stages.items():
    wrap(c, n)
N = 20
for i in range(3):
    for p in m.parameters(): p.grad = None
    m(**b)["loss"].backward()
torch.cuda.synchronize(); acc.clear()
tz = tf = tb = 0.0
t0 = time.perf_counter()
for i in range(N):
    t = time.perf_counter()
    for p in m.parameters(): p.grad = None
    tz += time.perf_counter() - t; t = time.perf_counter()
    out = m(**b)
    tf += time.perf_counter() - t; t = time.perf_counter()
    out["loss"].backward()
    tb += time.perf_counter() - t
th = time.perf_counter() - t0
torch.cuda.synchronize()
td = time.perf_counter() - t0
print(f"per step: host {th / N * 1e3:.2f} ms (zero grads {tz / N * 1e3:.2f}, forward call {tf / N * 1e3:.2f}, backward call {tb / N * 1e3:.2f}); device-complete {td / N * 1e3:.2f} ms")
for k in sorted(acc, key=lambda k: -acc[k]):
    print(f"  {k:32s} {acc[k] / N * 1e3:7.3f} ms/step")
