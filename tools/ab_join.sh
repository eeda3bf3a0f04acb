# A/B: deferred join of the weight-gradient stream (PENEO_DEFER_JOIN) — run through gpurun
cd $GRAFT_REPO_ROOT
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for v in 1 0 1 0; do
  echo "PENEO_DEFER_JOIN=$v"; PENEO_DEFER_JOIN=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['forward_only']['ms_per_batch'])"
done
