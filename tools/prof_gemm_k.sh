#!/bin/bash
# kernel durations (no launch gaps) of the QKV-shaped forward GEMM over K -> gpurun_out/gemmk/durations.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/gemmk; rm -rf $OUT; mkdir -p $OUT
cat > $OUT/run.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from peneo_amd import ops
M, N = 5672, 2304
for K in (64, 128, 256, 512, 768, 1536):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(10): ops.gemm(a, w, bias=bias, out=out)
    torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o run -- python3 $OUT/run.py > $OUT/log.txt 2>&1
T=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python - "$T" > $OUT/durations.txt <<'PY'
import csv, sys
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
g = [r for r in rows if "gemm" in r[2]]
for i in range(0, len(g), 10):
    grp = g[i:i + 10]
    d = sorted((e - s) / 1e3 for s, e, _ in grp)
    gaps = sorted((grp[k + 1][0] - grp[k][1]) / 1e3 for k in range(len(grp) - 1))
    print(f"{grp[0][2][:70]:70s} median {d[len(d) // 2]:6.1f} us  min {d[0]:6.1f}  gap median {gaps[len(gaps) // 2]:5.1f} us")
PY
rm -rf $OUT/prof; cat $OUT/durations.txt
