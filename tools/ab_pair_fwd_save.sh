# ablations of the saving forward (run through gpurun): variant libraries with -DPH_ABLATE=n (1 no record stores, 2 no x stores)
LIST="${LIST:-1 2 3}"
cd $GRAFT_REPO_ROOT/peneo_amd/csrc
for n in $LIST; do
  mkdir -p /tmp/ph$n
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DPH_ABLATE=$n -c pair_heads.hip -o /tmp/ph$n/pair_heads.o &
done
wait
cd $GRAFT_REPO_ROOT
echo "== default"; TIME=1 python tools/check_pair_saved.py 2>&1 | grep "forward"
for n in $LIST; do
  objs=$(ls peneo_amd/lib/obj/*.o | grep -v pair_heads.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/ph$n/lib.so $objs /tmp/ph$n/pair_heads.o
  echo "== PH_ABLATE=$n"; PENEO_HIP_LIB=/tmp/ph$n/lib.so python tools/check_pair_saved.py 2>&1 | grep "forward"
done
