"""Does a persistent one-workgroup-per-CU GEMM on a SIDE stream help or hurt the tiled kernels of the main stream?  (round 6)
Main stream: the four dgrad GEMMs of an encoder layer (the model's layouts), 12 layers.  Side stream: four weight-gradient-sized
GEMMs per layer (same FLOPs as dW = dy^T x, K = 5696 tokens, as k-major operands so that both kernel families can run them):
(a) the grouped tiled launch (432 tiles of 128 x 128, what the model runs), (b) four persistent stream-k launches of 256 x 128
tiles on G workgroups.  Reported: each stream alone, and the makespan of both together."""
import os, sys, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peneo_amd import ops, hip
hip.load_library()
lib = ctypes.CDLL(hip.LIB_PATH)
dev = torch.device("cuda:0")
bf = torch.bfloat16
R, H, I, K = 5672, 768, 3072, 5696
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(bf)
# main: dgrads  d_zi = d2 Wo2 [R,H]x[H,I]; d_a = d_zi Wi [R,I]x[I,H]; d_att = d1 Wo [R,H]x[H,H]; d_x = dqkv Wqkv [R,3H]x[3H,H]
d2, Wo2, dzi, Wi, d1, Wo, dqkv, Wqkv = rnd(R, H), rnd(H, I), rnd(R, I), rnd(I, H), rnd(R, H), rnd(H, H), rnd(R, 3 * H), rnd(3 * H, H)
o_zi, o_a, o_att, o_x = (torch.empty(R, n, device=dev, dtype=bf) for n in (I, H, H, H))
def main_layer():
    ops.gemm(d2, Wo2, b_kmajor=False, out=o_zi, split_k=1)
    ops.gemm(dzi, Wi, b_kmajor=False, out=o_a, split_k=1)
    ops.gemm(d1, Wo, b_kmajor=False, out=o_att, split_k=1)
    ops.gemm(dqkv, Wqkv, b_kmajor=False, out=o_x, split_k=1)
# side: wgrad-sized problems [M, K] x [N, K]^T -> fp32 [M, N]
shapes = [(3 * H, H), (I, H), (H, I), (H, H)]
sa = [rnd(m, K) for m, n in shapes]; sb = [rnd(n, K) for m, n in shapes]
so = [torch.empty(m, n, device=dev, dtype=torch.float32) for m, n in shapes]
def side_tiled():
    ops.gemm_group([(a, b, o) for a, b, o in zip(sa, sb, so)], a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)
def side_sk(mode):
    lib.peneo_gemm_set_sk_mode(mode)
    for a, b, o in zip(sa, sb, so):
        ops.gemm(a, b, out=o, split_k=1)
    lib.peneo_gemm_set_sk_mode(0)
s_main, s_side = torch.cuda.Stream(), torch.cuda.Stream()
def run(main_on, side_fn, n=12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if main_on:
            with torch.cuda.stream(s_main): main_layer()
        if side_fn:
            with torch.cuda.stream(s_side): side_fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
def best(main_on, side_fn):
    run(main_on, side_fn, 3)
    return min(run(main_on, side_fn) for _ in range(5)) / 12 * 1e3
lib.peneo_gemm_set_sk_mode(0)
print(f"us per layer: main alone {best(True, None):7.1f}")
print(f"              side alone, tiled group {best(False, side_tiled):7.1f}    both {best(True, side_tiled):7.1f}")
for mode in (108128, 108256):
    for G in (0, 192, 128, 96, 64):
        lib.peneo_gemm_sk_set_max_groups(G)
        f = lambda: side_sk(mode)
        print(f"              side alone, stream-k {mode} on {G or 256:3d} workgroups {best(False, f):7.1f}    both {best(True, f):7.1f}")
lib.peneo_gemm_sk_set_max_groups(0)
