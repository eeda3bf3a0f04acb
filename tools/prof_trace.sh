# kernel trace (csv) of a short bench run -> gpurun_out/$1 ; run through gpurun:  bash tools/prof_trace.sh NAME [bench args]
NAME=${1:-trace}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$NAME -o run -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/$NAME.log 2>&1
find gpurun_out/$NAME -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/timeline.py {} 2 > gpurun_out/$NAME.timeline.txt 2>&1
head -50 gpurun_out/$NAME.timeline.txt
