#!/bin/bash
# kernel table of LiLT-base train steps (BASELINE config 5) -> gpurun_out/lilt/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/lilt; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o run -- python3 bench.py --backbone lilt --steps 5 --warmup 2 --no-cpu-baseline --no-ragged > $OUT/line.json 2> $OUT/err.txt
T=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python tools/prof_summary_csv.py $T 30 > $OUT/summary.txt 2>&1
rm -rf $OUT/prof; head -36 $OUT/summary.txt
