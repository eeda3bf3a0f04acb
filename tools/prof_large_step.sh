#!/bin/bash
# kernel table of LayoutLMv3-large train steps (BASELINE config 4: S = 1024, 2 documents) -> gpurun_out/large/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/large; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o run -- python3 bench.py --size large --seq-len 1024 --lines 256 --docs-per-gpu 2 --steps 5 --warmup 2 --no-cpu-baseline --no-ragged > $OUT/line.json 2> $OUT/err.txt
T=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python tools/prof_summary_csv.py $T 30 > $OUT/summary.txt 2>&1
rm -rf $OUT/prof; head -30 $OUT/summary.txt
