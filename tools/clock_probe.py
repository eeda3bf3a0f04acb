"""Shader clock a kernel actually runs at: s_memtime ticks of a tiny timing kernel are not available from Python, so this loops ONE operation for a
few seconds while rocm-smi samples sclk / power in a second process.   python tools/clock_probe.py pair_fwd | pair_bwd | gemm | attn"""
import math, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
what = sys.argv[1] if len(sys.argv) > 1 else "pair_fwd"
dt, dev = torch.bfloat16, "cuda"
B, N, D, classes, nh = 8, 511, 384, [2, 3, 3, 3, 3], 5
P = N * (N + 1) // 2
if what in ("pair_fwd", "pair_bwd"):
    ab = torch.randn(B, N, 2 * D, device=dev).to(dt)
    w1 = [torch.randn(D, D, device=dev) / math.sqrt(D) for _ in classes]
    w2 = [torch.randn(c, D, device=dev) / math.sqrt(D) for c in classes]
    b1, b2 = torch.zeros(nh * D, device=dev), torch.zeros(14, device=dev)
if what == "pair_fwd":
    wp = ops.pair_heads_pack(dt, w1, w2)
    tags = [torch.zeros(B, P, dtype=torch.int64, device=dev) for _ in classes]; cw = [torch.ones(c, device=dev) for c in classes]
    fn = lambda: ops.pair_heads_fwd(ab, wp, b1, b2, classes, tags=tags, class_weights=cw, want_dlogits=True, want_logits=False, drop_p=0.1, drop_seed=1)
elif what == "pair_bwd":
    dl = [torch.randn(B, P, c, device=dev) * 1e-3 for c in classes]
    wp2 = ops.pair_bwd_pack(w1); rows = ops.pair_bwd_rows(N)
    dz = torch.empty(B * rows, nh * D, device=dev, dtype=dt); x = torch.empty(B * rows, D, device=dev, dtype=dt)
    d_ab = torch.zeros(B, N, 2 * D, device=dev); ws = ops.pair_dz_workspace(nh, D, dev, slots=256)
    args = ops.pair_dz_args(D, classes, dl, w2, torch.ones(nh, device=dev), drop_p=0.1, drop_seed=7)
    fn = lambda: ops.pair_bwd_fused(ab, wp2, b1, args, dz, x, d_ab, ws)
elif what == "gemm":
    a = torch.randn(5672, 768, device=dev).to(dt); w = torch.randn(3072, 768, device=dev).to(dt)
    fn = lambda: ops.gemm(a, w)
else:
    a = torch.randn(8192, 8192, device=dev).to(dt)
    fn = lambda: a.add_(1.0)
for _ in range(3): fn()
torch.cuda.synchronize()
mon = subprocess.Popen("for i in 1 2 3 4 5 6; do sleep 0.45; rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | tr -s ' ' | head -3 | tr '\\n' ' '; echo; done", shell=True)
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < 3.2:
    for _ in range(20): fn()
    n += 20
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
mon.wait()
print(f"{what}: {e0.elapsed_time(e1) / n * 1e3:8.1f} us per launch over {n} launches")
