# Sweep of one environment switch over several values, two rounds: bash tools/ab_sweep.sh VAR v1 v2 ...   (run through gpurun)
VAR=$1; shift
cd $GRAFT_REPO_ROOT
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for round in 1 2; do
  for v in "$@"; do
    echo -n "$VAR=$v  "; env $VAR=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['forward_only']['ms_per_batch'])"
  done
done
