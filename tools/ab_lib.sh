# A/B of two builds of the library on one box: peneo_amd/lib/libpeneo_old.so (copy of the previous build) against
# libpeneo_hip.so; bash tools/ab_lib.sh [script.py ...]   (run through gpurun)
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for lib in libpeneo_old.so libpeneo_hip.so; do
  for s in "$@"; do
    echo "== $lib $s"
    PENEO_HIP_LIB=$GRAFT_REPO_ROOT/peneo_amd/lib/$lib python $s 2>&1 | grep -v amdgpu.ids
  done
done
done
for r in 1 2 3; do
for lib in libpeneo_old.so libpeneo_hip.so; do
  echo -n "== $lib  "
  PENEO_HIP_LIB=$GRAFT_REPO_ROOT/peneo_amd/lib/$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['forward_only']['ms_per_batch'])"
done
done
