"""embed_text_bwd with subsets of its tables (which gradients make the token-per-wave atomics slow?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
from peneo_amd.data import synthetic_rfund_batch
B, S, H, V = 8, 512, 768, 50265
b = {k: v.cuda() for k, v in synthetic_rfund_batch(B, S, 128, V, seed=1, with_image=False).items()}
ids, bbox = b["input_ids"], b["bbox"]
pid = ops.position_ids(ids, 1)
d = torch.randn(B, S, H, device="cuda").to(torch.bfloat16)
g = dict(word=torch.zeros(V, H, device="cuda"), pos=torch.zeros(514, H, device="cuda"), x=torch.zeros(1024, 128, device="cuda"),
         y=torch.zeros(1024, 128, device="cuda"), h=torch.zeros(1024, 128, device="cuda"), w=torch.zeros(1024, 128, device="cuda"))
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
full = lambda: ops.embed_bwd(d, B, S, H, input_ids=ids, pos_ids=pid, bbox=bbox, g_word=g["word"], g_pos=g["pos"], g_x=g["x"], g_y=g["y"], g_h=g["h"], g_w=g["w"])
text = lambda: ops.embed_bwd(d, B, S, H, input_ids=ids, pos_ids=pid, g_word=g["word"], g_pos=g["pos"])
spat = lambda: ops.embed_bwd(d, B, S, H, bbox=bbox, g_x=g["x"], g_y=g["y"], g_h=g["h"], g_w=g["w"])
print(f"all tables {bench(full):7.1f} us   word + pos {bench(text):7.1f} us   box tables {bench(spat):7.1f} us")
