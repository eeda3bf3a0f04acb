# A/B of the decoder-backward options (run through gpurun)
for cfg in "2" "1" "2" "1"; do set -- $cfg
  echo -n "DEC_STREAMS=$1: "
  PENEO_DEC_STREAMS=$1 timeout 300 python tools/run_phases.py 2>&1 | tail -3 | tr '\n' ' '
  PENEO_DEC_STREAMS=$1 timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c60-85
done
