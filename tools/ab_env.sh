# A/B of one environment switch: bash tools/ab_env.sh VAR valA valB [bench args]   (run through gpurun)
VAR=$1; A=$2; B=$3; shift 3
cd $GRAFT_REPO_ROOT
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for v in $A $B $A $B; do
  echo -n "$VAR=$v  "; env $VAR=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['forward_only']['ms_per_batch'])"
done
