#!/bin/bash
# per-kernel durations of the attention micro-benchmark (tools/run_attn.py) -> gpurun_out/$1/attn_kernels.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-attn}; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o run -- python3 tools/run_attn.py > $OUT/run.log 2>&1
S=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
python - "$S" > $OUT/attn_kernels.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "attn" in n or "drop_words" in n:
        print(f'{float(r["AverageNs"]) / 1e3:9.1f} us  x{r["Calls"]:>4}  {n[:110]}')
PY
rm -rf $OUT/prof
cat $OUT/attn_kernels.txt
