#!/bin/bash
# kernel-trace of 5 bench steps -> per-kernel table + step timeline (gpurun_out/$1/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-prof}; rm -rf $OUT; mkdir -p $OUT
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline ${@:2} > $OUT/steps5_line.json 2> $OUT/steps5.err
T=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T +4 --gaps > $OUT/step_timeline.txt 2>&1
python tools/timeline.py $T +4 --list > $OUT/step_list.txt 2>&1
python tools/prof_summary_csv.py $T 45 > $OUT/steps5_summary.txt 2>&1
rm -rf $OUT/prof
cat $OUT/steps5_summary.txt | head -50
