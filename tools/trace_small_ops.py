"""Which host lines launch torch-native kernels inside one benchmark step?  CPU-side torch.profiler with stacks: every aten op
that reaches a device kernel (fill_, copy_, mul, cat, neg, arange, add, clone ...) grouped by the innermost peneo_amd / bench frame."""
import sys, os, re, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from seeded import layoutlmv3_config, peneo_config
from peneo_amd.model import PEneoConfig, PEneoModel
from peneo_amd.data import synthetic_rfund_batch
from torch.profiler import profile, ProfilerActivity

pcfg = peneo_config("layoutlmv3-base", layoutlmv3_config("base"))
m = PEneoModel(PEneoConfig(**{k: v for k, v in pcfg.items() if k != "model_type"})).cuda().set_compute_dtype(torch.bfloat16).train()
m.backbone.check_inputs = False
b = {k: v.cuda() for k, v in synthetic_rfund_batch(8, 512, 128, 50265, seed=1).items()}

EVAL = os.environ.get("EVAL", "0") == "1"      # EVAL=1: the eval forward alone
if EVAL:
    m.eval()


def step():
    if EVAL:
        with torch.no_grad():
            m(**b)
        return
    for p in m.parameters():
        p.grad = None
    m(**b)["loss"].backward()

for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step()
    torch.cuda.synchronize()
LEAF = {"aten::fill_", "aten::zero_", "aten::copy_", "aten::mul", "aten::mul_", "aten::cat", "aten::neg", "aten::arange", "aten::add", "aten::add_",
        "aten::sub", "aten::div", "aten::sum", "aten::stack", "aten::index_select", "aten::where", "aten::eq", "aten::ne", "aten::cumsum", "aten::_to_copy",
        "aten::clone", "aten::contiguous", "aten::zeros", "aten::zeros_like", "aten::full", "aten::ones", "aten::sqrt", "aten::rsqrt", "aten::exp",
        "aten::masked_fill_", "aten::masked_fill", "aten::bitwise_and", "aten::lt", "aten::gt", "aten::ge", "aten::le", "aten::index", "aten::gather"}
KERNEL = {"aten::zero_", "aten::fill_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::empty_like", "aten::expand", "aten::reshape", "aten::copy_", "aten::mul", "aten::mul_", "aten::cat", "aten::neg", "aten::arange", "aten::add", "aten::add_", "aten::sub",
          "aten::div", "aten::sum", "aten::index_select", "aten::where", "aten::eq", "aten::ne", "aten::cumsum", "aten::masked_fill_", "aten::index",
          "aten::gather", "aten::lt", "aten::gt", "aten::ge", "aten::le", "aten::bitwise_and", "aten::sqrt", "aten::rsqrt", "aten::exp"}
cnt = collections.Counter()
for e in prof.events():
    if e.name not in KERNEL:
        continue
    where = "?"
    frames = [fr for fr in e.stack if "peneo_amd" in fr or "trace_small_ops" in fr]
    if frames:
        where = " <- ".join(re.sub(r"^.*/(peneo_amd|tools)/", "", fr) for fr in frames[:3])
    elif e.stack:
        where = "|".join(fr[-50:] for fr in e.stack[:3])
    shapes = ""
    cnt[(e.name, where)] += 1
tot = 0
for (name, where), c in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(f"{c:4d}  {name:22s} {where}")
    tot += c
print("total", tot)
