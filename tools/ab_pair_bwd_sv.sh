# ablation ladder of peneo_pair_bwd_saved (run through gpurun): variant libraries with -DPB_ABLATE=n
# (saved-activation kernel: 1 no dz row stores, 2 no du MFMAs, 8 no exp / rcp, 16 no record DMA, 32 no weight DMA)
LIST="${LIST:-a1 a2 a8 a16 a32 a59}"   # e.g. a0_r6_s1 = no ablation, six record slots, spread weight pieces
cd $GRAFT_REPO_ROOT/peneo_amd/csrc
for n in $LIST; do
  mkdir -p /tmp/pb$n
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 $(echo $n | sed "s/^a/-DPB_ABLATE=/; s/_r/ -DPSV_NR=/; s/_s/ -DPSV_SPREAD=/") -c pair_bwd.hip -o /tmp/pb$n/pair_bwd.o 2>/dev/null &
done
wait
cd $GRAFT_REPO_ROOT
echo "== default"; python tools/check_pair_saved.py 2>&1 | grep "backward, saved" | tail -1
for n in $LIST; do
  objs=$(ls peneo_amd/lib/obj/*.o | grep -v pair_bwd.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/pb$n/lib.so $objs /tmp/pb$n/pair_bwd.o
  echo "== PB_ABLATE=$n"; PENEO_HIP_LIB=/tmp/pb$n/lib.so python tools/check_pair_saved.py 2>&1 | grep "backward, saved" | tail -1
done
