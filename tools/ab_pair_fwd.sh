# ablations of the hand-interleaved pair forward (run through gpurun): variant libraries with -DPH_ABLATE=n (8: no exp / rcp), eval and train launches
LIST="${LIST:-8}"
cd $GRAFT_REPO_ROOT/peneo_amd/csrc
for n in $LIST; do
  mkdir -p /tmp/ph$n
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DPH_ABLATE=$n -c pair_heads.hip -o /tmp/ph$n/pair_heads.o &
done
wait
cd $GRAFT_REPO_ROOT
for r in 1 2; do
echo "== default"; TRAIN=0 python tools/run_pair.py | tail -1; python tools/run_pair.py | tail -1
for n in $LIST; do
  objs=$(ls peneo_amd/lib/obj/*.o | grep -v pair_heads.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/ph$n/lib.so $objs /tmp/ph$n/pair_heads.o
  echo "== PH_ABLATE=$n"; TRAIN=0 PENEO_HIP_LIB=/tmp/ph$n/lib.so python tools/run_pair.py | tail -1; PENEO_HIP_LIB=/tmp/ph$n/lib.so python tools/run_pair.py | tail -1
done; done
