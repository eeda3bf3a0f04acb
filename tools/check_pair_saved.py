"""The saving forward + peneo_pair_bwd_saved against the plain forward + peneo_pair_bwd_fused on the same inputs: logits / dlogits / x rows must be
identical (same arithmetic per pair, another walk), dz / d_ab / the dW2, db1 sums agree to the rounding of the saved factors; then timings.
python tools/check_pair_saved.py [B N]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops
B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 150)
D, classes, nh = 384, [2, 3, 3, 3, 3], 5
dt, dev = torch.bfloat16, "cuda"
torch.manual_seed(3)
P = N * (N + 1) // 2
bad = 0
for drop in (0.1, 0.0):
    ab = torch.randn(B, N, 2 * D, device=dev).to(dt)
    w1 = [torch.randn(D, D, device=dev) / math.sqrt(D) for _ in classes]
    w2 = [torch.randn(c, D, device=dev) / math.sqrt(D) for c in classes]
    b1, b2 = 0.1 * torch.randn(nh * D, device=dev), 0.1 * torch.randn(14, device=dev)
    wp = ops.pair_heads_pack(dt, w1, w2)
    tags = [torch.randint(0, c, (B, P), device=dev) for c in classes]
    cw = [torch.rand(c, device=dev) + 0.5 for c in classes]
    kw = dict(tags=tags, class_weights=cw, want_dlogits=True, want_logits=True, drop_p=drop, drop_seed=77)
    lg0, pt0, dl0 = ops.pair_heads_fwd(ab, wp, b1, b2, classes, **kw)
    lg1, pt1, dl1, (act, xr) = ops.pair_heads_fwd(ab, wp, b1, b2, classes, save=True, **kw)
    torch.cuda.synchronize()
    line = f"drop={drop} B={B} N={N}: "
    for h in range(nh):
        nd = int((lg0[h] != lg1[h]).sum()) + int((dl0[h] != dl1[h]).sum())
        bad += nd
        line += f"head {h} logits+dlogits differ {nd}; "
    s0, s1 = pt0.sum(0), pt1.sum(0)
    rel = float((s0 - s1).abs().max() / s0.abs().max())
    line += f"loss partial sums rel {rel:.1e}"
    bad += rel > 1e-4
    print(line, flush=True)
    # backward
    wp2 = ops.pair_bwd_pack(w1); rows = ops.pair_bwd_rows(N)
    scale = torch.rand(nh, device=dev) + 0.5
    outs = []
    for saved in (False, True):
        dz = torch.full((B * rows, nh * D), 3.0, device=dev, dtype=dt)
        d_ab = torch.zeros(B, N, 2 * D, device=dev)
        ws = ops.pair_dz_workspace(nh, D, dev, slots=256)
        args = ops.pair_dz_args(D, classes, dl0, w2, scale, drop_p=drop, drop_seed=77)
        if saved:
            ops.pair_bwd_saved(ab, wp2, args, act, dz, d_ab, ws)
            x = xr
        else:
            x = torch.empty(B * rows, D, device=dev, dtype=dt)
            ops.pair_bwd_fused(ab, wp2, b1, args, dz, x, d_ab, ws)
        torch.cuda.synchronize()
        outs.append((dz.float(), x.float(), d_ab, ws.sum(0)))
    names = ["dz", "x", "d_ab", "sums"]
    line = "   backward: "
    for i, n in enumerate(names):
        a, b_ = outs[0][i], outs[1][i]
        rel = float((a - b_).norm() / (a.norm() + 1e-30))
        line += f"{n} rel {rel:.2e} (max abs {float((a - b_).abs().max()):.2e} of {float(a.abs().max()):.2e}); "
        bad += (rel > (0 if n == "x" else 4e-3)) or not torch.isfinite(b_).all()
    print(line, flush=True)
print("FAILED" if bad else "all close")

if os.environ.get("TIME", "1") == "1":
    B, N = 8, 511
    P = N * (N + 1) // 2
    ab = torch.randn(B, N, 2 * D, device=dev).to(dt)
    w1 = [torch.randn(D, D, device=dev) / math.sqrt(D) for _ in classes]
    w2 = [torch.randn(c, D, device=dev) / math.sqrt(D) for c in classes]
    b1, b2 = torch.zeros(nh * D, device=dev), torch.zeros(14, device=dev)
    wp = ops.pair_heads_pack(dt, w1, w2); wp2 = ops.pair_bwd_pack(w1); rows = ops.pair_bwd_rows(N)
    tags = [torch.zeros(B, P, dtype=torch.int64, device=dev) for _ in classes]; cw = [torch.ones(c, device=dev) for c in classes]
    kw = dict(tags=tags, class_weights=cw, want_dlogits=True, want_logits=False, drop_p=0.1, drop_seed=1234)
    _, _, dl, (act, xr) = ops.pair_heads_fwd(ab, wp, b1, b2, classes, save=True, **kw)
    dz = torch.empty((B * rows, nh * D), device=dev, dtype=dt); x = torch.empty(B * rows, D, device=dev, dtype=dt)
    d_ab = torch.zeros(B, N, 2 * D, device=dev); ws = ops.pair_dz_workspace(nh, D, dev, slots=256)
    args = ops.pair_dz_args(D, classes, dl, w2, torch.ones(nh, device=dev), drop_p=0.1, drop_seed=1234)
    def bench(name, fn, n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{name:44s} {e0.elapsed_time(e1) / n * 1e3:8.1f} us", flush=True)
    for rep in range(2):
        bench("forward (train), plain walk", lambda: ops.pair_heads_fwd(ab, wp, b1, b2, classes, **kw))
        bench("forward (train), saving", lambda: ops.pair_heads_fwd(ab, wp, b1, b2, classes, save=True, **kw))
        bench("backward, fused (recompute)", lambda: ops.pair_bwd_fused(ab, wp2, b1, args, dz, x, d_ab, ws))
        bench("backward, saved activations", lambda: ops.pair_bwd_saved(ab, wp2, args, act, dz, d_ab, ws))
