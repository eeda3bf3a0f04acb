LIST="${LIST:-65536}"
cd $GRAFT_REPO_ROOT/peneo_amd/csrc
for n in $LIST; do
  mkdir -p /tmp/pd$n
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -DPB_DBG=$n -c pair_bwd.hip -o /tmp/pd$n/pair_bwd.o &
done
wait
cd $GRAFT_REPO_ROOT
PENEO_PB_MODE=1w python tools/dbg_pb3.py save 2>&1 | grep -v amdgpu
for n in $LIST; do
  objs=$(ls peneo_amd/lib/obj/*.o | grep -v pair_bwd.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/pd$n/lib.so $objs /tmp/pd$n/pair_bwd.o
  echo "PB_DBG=$n: "; PENEO_HIP_LIB=/tmp/pd$n/lib.so python tools/dbg_pb3.py cmp 2>&1 | grep -v amdgpu | cut -c1-200 | head -8
  /opt/rocm/lib/llvm/bin/llvm-objdump -d --offloading /tmp/pd$n/pair_bwd.o 2>/dev/null | grep -c v_pk_ 
done
