# dW1 side GEMM: split-k x join policy (run through gpurun)
cd $GRAFT_REPO_ROOT
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for round in 1 2; do
  for cfg in "0 0" "0 1" "5 1" "4 1" "3 1" "7 1"; do
    set -- $cfg
    echo -n "split=$1 hold=$2  "; env PENEO_DW1_SPLIT=$1 PENEO_DW1_HOLD=$2 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['forward_only']['ms_per_batch'])"
  done
done
