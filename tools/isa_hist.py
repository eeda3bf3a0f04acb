#!/usr/bin/env python3
"""Instruction histogram of a kernel's ISA between its s_barriers (what a LONE wave pays: one issue slot of ~4 cycles per
instruction of any kind).  Text order, not execution order: meaningful for straight-line loop bodies (the one-wave pair backward).
    python tools/isa_hist.py pair_bwd.hip pair_bwd_one_kernelILi32ELb1E [segment index, default: the longest]"""
import os, re, subprocess, sys, tempfile
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "peneo_amd", "csrc", sys.argv[1])
want = sys.argv[2]
extra = ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1"] if "pair_bwd" in src else []
out = os.path.join(tempfile.mkdtemp(), "k.s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S", src, "-o", out] + extra,
               check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
names = [n for n in re.findall(r"^(_Z\w+):", s, re.M) if want in n]
if not names:
    sys.exit(f"no kernel symbol contains {want!r}")
start = s.index(names[0] + ":"); end = s.index("s_endpgm", start)
ins = [l.strip() for l in s[start:end].splitlines()]
ins = [l for l in ins if l and not l.startswith((";", ".", "//")) and not l.endswith(":")]
bars = [i for i, l in enumerate(ins) if l.startswith("s_barrier")]
segs = list(zip([0] + bars, bars + [len(ins)]))
print(f"{names[0]}: {len(ins)} instructions, s_barrier at {bars}")
k = int(sys.argv[3]) if len(sys.argv) > 3 else max(range(len(segs) - 1), key=lambda i: (segs[i][1] - segs[i][0]) if any("v_mfma" in l for l in ins[segs[i][0]:segs[i][1]]) else 0)
a, b = segs[k]
c = Counter(l.split()[0] for l in ins[a:b])
print(f"segment {k}: {b - a} instructions")
print(", ".join(f"{op}:{n}" for op, n in c.most_common()))
