#!/bin/bash
# HBM-side kernel table of one serialised train step (gpurun_out/$1/hbm_kernels.txt): bash tools/prof_hbm.sh [outdir] [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-hbm}; rm -rf $OUT; mkdir -p $OUT
export PENEO_DEC_STREAMS=1 PENEO_WGRAD_STREAM=0 PENEO_DW1_SIDE=0
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ragged > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ragged ${@:2} > $OUT/line.json 2> $OUT/err.txt
T=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
head -1 $T > $OUT/csv_header.txt
python tools/hbm_table.py $T > $OUT/hbm_kernels.txt 2>&1
rm -rf $OUT/prof
cat $OUT/hbm_kernels.txt
