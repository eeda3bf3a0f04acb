# A/B of the decoder-backward stream assignment and of the fused-dz kernel's software pipeline (run through gpurun)
for cfg in "0 1 1" "0 1 0" "1 0 1" "0 0 1"; do set -- $cfg
  echo -n "X_SIDE=$1 DW_MAIN=$2 DZF_PIPE=$3: "
  PENEO_DZ_X_SIDE=$1 PENEO_DZ_DW_MAIN=$2 PENEO_DZF_PIPE=$3 timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c60-85
done
echo -n "unfused: "; PENEO_DZ_FUSED=0 timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c60-85
