# Reproduces the packed-fp32 corruption of the wave-specialised pair_bwd kernel (run through gpurun): builds the kernel with
# v_pk_*_f32 dz arithmetic (-DPB_DBG=65536) and compares dz with the one-wave-per-SIMD kernel's, six launches.
cd $GRAFT_REPO_ROOT/peneo_amd/csrc
mkdir -p /tmp/pd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DPB_DBG=65536 -c pair_bwd.hip -o /tmp/pd/pair_bwd.o
cd $GRAFT_REPO_ROOT
objs=$(ls peneo_amd/lib/obj/*.o | grep -v pair_bwd.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/pd/lib.so $objs /tmp/pd/pair_bwd.o
PENEO_PB_MODE=1w python tools/dbg_pb3.py save 2>&1 | grep -v amdgpu
echo "scalar arithmetic (shipped):"; python tools/dbg_pb3.py cmp 2>&1 | grep -v amdgpu | head -8
echo "packed arithmetic (-DPB_DBG=65536):"; PENEO_HIP_LIB=/tmp/pd/lib.so python tools/dbg_pb3.py cmp 2>&1 | grep -v amdgpu | head -12
