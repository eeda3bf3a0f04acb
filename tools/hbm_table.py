#!/usr/bin/env python3
"""HBM-side kernels of one train step against the 8 TB/s HBM3E peak (north_star: "coalesced HBM loads ... evidenced by rocprof
HBM GB/s"): per kernel the ALGORITHMIC bytes of one launch (what it must read + write once, from the shapes of BASELINE
config 2: B = 8 documents, S = 512, T = 709, H = 768, I = 3072, 12 heads, 12 layers) divided by its rocprofv3 duration in a
step whose streams are serialised (PENEO_WGRAD_STREAM=0 PENEO_DEC_STREAMS=1 PENEO_DW1_SIDE=0: no co-running kernel shares the
bandwidth).  Launches of one kernel name with different shapes are told apart by their grid size.

    python tools/hbm_table.py <kernel_trace.csv> [vocab=50265]"""
import csv, sys
from collections import defaultdict
B, S, T, H, I, NH, L = 8, 512, 709, 768, 3072, 12, 12
R, Tp = B * T, 768
vocab = int(sys.argv[2]) if len(sys.argv) > 2 else 50265
bf, f4 = 2, 4
rows = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows[r["Kernel_Name"]].append(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                       int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])))
def med(v):
    v = sorted(v); return v[len(v) // 2]
out = []
def add(label, match, nbytes, note="", pick=None):
    for name, v in rows.items():
        if match in name:
            d = [x for x, g in v]
            if pick is not None:       # keep the launches whose duration is within 2x of the largest cluster asked for
                d = pick(v)
            if not d: continue
            t = med(d)
            out.append((label, len(d), t, nbytes, nbytes / t / 1e3 if nbytes else None, note))
            return
big = lambda v: [x for x, g in v if g >= max(g2 for _, g2 in v) * 0.9]           # the largest-grid launches of a name
add("ln_fwd32 [5672, 768] bf16", "ln_fwd32_kernel<unsigned short, 3, false", 2 * R * H * bf, "x read, y written", big)
add("ln_bwd32 [5672, 768] (+ dropped 2nd output)", "ln_bwd32_kernel<unsigned short, 3, false", 4 * R * H * bf, "dy, x read; dx, dx_dropped written", big)
add("embed_text_fwd 4096 tokens", "embed_text_fwd_kernel", B * S * (6 * H * f4 + H * bf), "6 table rows of fp32 per token read, bf16 row written")
add("embed_text_bwd (word / position scatter)", "embed_text_bwd_kernel", B * S * (H * bf + 2 * 2 * H * f4), "d_x row read, 2 fp32 rows read-modify-written (atomics)")
add("embed_box_bwd (4 + 2 box tables)", "embed_box_bwd_kernel", B * S * (H * bf + 2 * H * f4), "d_x row read, H fp32 values RMW")
add("relpos_bias_fwd [8, 12, 709, 768] bf16", "relpos_bias_fwd_kernel", B * NH * T * Tp * bf + 3 * B * T * T, "3 u8 bucket maps read, bias written")
add("relpos_bias_bwd_layers (12 dS^T slabs)", "relpos_bias_bwd_layers_kernel", L * B * NH * T * Tp * bf + 3 * B * T * Tp, "slabs read once")
add("attn_drop_words (12 layers)", "attn_drop_words_kernel", L * B * NH * 24 * 768 * 4, "keep words written (integer-bound generator)")
add("colsum [5672, 3072] bf16", "colsum_vec_kernel<unsigned short>", R * I * bf, "read once", big)
add("splitk_reduce8 (largest)", "splitk_reduce8_kernel", None, "")
add("copy_rows (crop)", "copy_rows_kernel", 2 * B * S * H * bf, "")
add("adamw (127 M parameters)", "adamw_kernel", 127.25e6 * 7 * f4, "p, g, m, v read; p, m, v written")
add("grad_sqnorm (127 M gradients)", "grad_sqnorm_kernel", 127.25e6 * f4, "read once")
add("cast_multi (85 M encoder weights)", "cast_multi_kernel", 85e6 * (f4 + bf), "fp32 read, bf16 written")
add("pair_bwd_reduce", "pair_bwd_reduce_kernel", None, "")
print(f"{'kernel (shape)':46s} {'calls':>5s} {'median us':>10s} {'alg. MB':>9s} {'GB/s':>8s} {'of 8 TB/s':>9s}  traffic counted")
for label, n, t, nb, gbs, note in out:
    if nb is None:
        print(f"{label:46s} {n:5d} {t:10.1f} {'':>9s} {'':>8s} {'':>9s}  {note}")
    else:
        print(f"{label:46s} {n:5d} {t:10.1f} {nb / 1e6:9.1f} {gbs:8.0f} {100 * gbs / 8000:8.1f}%  {note}")
