#!/bin/bash
# Instrumented variants of the library next to the product one (per-phase s_memtime counters, compiled out of the product build):
#   bash tools/prof_build.sh attn    -> peneo_amd/lib/libpeneo_attnprof.so  (attention.hip -DATTN_PROF;  tools/attn_cycles.py)
#   bash tools/prof_build.sh pairbwd -> peneo_amd/lib/libpeneo_pbprof.so    (pair_bwd.hip -DPB_PROF;     tools/pb_cycles.py)
#   bash tools/prof_build.sh pairfwd -> peneo_amd/lib/libpeneo_phprof.so    (pair_heads.hip -DPH_PROF;   tools/ph_cycles.py)
# Build here (CPU box), run through gpurun with PENEO_HIP_LIB=$PWD/peneo_amd/lib/<that library>; delete the file afterwards (it travels
# with every gpurun push).
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
mkdir -p /tmp/peneo_prof
case "$1" in
  attn)    src=attention; def=-DATTN_PROF; out=libpeneo_attnprof.so; extra="" ;;
  pairbwd) src=pair_bwd;  def=-DPB_PROF;   out=libpeneo_pbprof.so;   extra="-fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1" ;;
  pairfwd) src=pair_heads; def=-DPH_PROF;  out=libpeneo_phprof.so;   extra="" ;;
  *) echo "usage: $0 attn|pairbwd|pairfwd"; exit 2 ;;
esac
/opt/rocm/bin/hipcc $FLAGS $extra $def -c peneo_amd/csrc/$src.hip -o /tmp/peneo_prof/$src.o || exit 1
objs=$(ls peneo_amd/lib/obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o peneo_amd/lib/$out $objs /tmp/peneo_prof/$src.o && echo built peneo_amd/lib/$out
