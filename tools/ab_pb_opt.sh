# variant builds of pair_bwd.hip by -DPB_OPT=n (run through gpurun)
cd $GRAFT_REPO_ROOT/peneo_amd/csrc
for n in 0 1 2 3; do
  mkdir -p /tmp/po$n
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DPB_OPT=$n -c pair_bwd.hip -o /tmp/po$n/pair_bwd.o &
done
wait
cd $GRAFT_REPO_ROOT
for n in 0 1 2 3; do
  objs=$(ls peneo_amd/lib/obj/*.o | grep -v pair_bwd.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/po$n/lib.so $objs /tmp/po$n/pair_bwd.o
  echo "== PB_OPT=$n"; PENEO_HIP_LIB=/tmp/po$n/lib.so python tools/dbg_pb.py 2>&1 | grep -v amdgpu.ids | head -4; PENEO_HIP_LIB=/tmp/po$n/lib.so python tools/run_pair_bwd.py 2>&1 | grep pair_bwd_fused
done
