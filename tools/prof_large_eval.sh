#!/bin/bash
# kernel table of the large-backbone eval forward (config 4) with the persistent GEMM rules on / off
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for SK in 1 0; do
  OUT=gpurun_out/largeprof_sk$SK; rm -rf $OUT; mkdir -p $OUT
  export PENEO_GEMM_SK=$SK SIZE=large SEQ=1024 LINES=256 DOCS=2
  rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o run -- python3 tools/run_eval_fwd.py > $OUT/line.txt 2>&1
  T=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
  python tools/prof_summary_csv.py $T 14 > $OUT/summary.txt 2>&1
  rm -rf $OUT/prof; echo "== PENEO_GEMM_SK=$SK"; cat $OUT/line.txt | tail -2; head -18 $OUT/summary.txt | cut -c1-180
done
