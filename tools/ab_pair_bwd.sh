# ablation ladder of pair_bwd_fused (run through gpurun): builds variant libraries with -DPB_ABLATE=n
LIST="${LIST:-1 16 32 33}"
cd $GRAFT_REPO_ROOT/peneo_amd/csrc
for n in $LIST; do
  mkdir -p /tmp/pb$n
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -DPB_ABLATE=$n -c pair_bwd.hip -o /tmp/pb$n/pair_bwd.o &
done
wait
cd $GRAFT_REPO_ROOT
echo "== default"; python tools/run_pair_bwd.py 2>&1 | grep -v amdgpu
for n in $LIST; do
  objs=$(ls peneo_amd/lib/obj/*.o | grep -v pair_bwd.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/pb$n/lib.so $objs /tmp/pb$n/pair_bwd.o
  echo "== PB_ABLATE=$n (1 no dz stores, 2 no du MFMA, 4 no z MFMA, 8 no epilogue, 16 no du DMA, 32 no DMA, 64 no z fragment reads, 128 no mask generation)"; PENEO_HIP_LIB=/tmp/pb$n/lib.so python tools/run_pair_bwd.py 2>&1 | grep pair_bwd_fused
done
