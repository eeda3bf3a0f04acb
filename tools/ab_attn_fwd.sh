#!/bin/bash
# forward-attention ablation ladder (tools/_abl_N.so built with -DATTN_ABLATE=N; see the hooks in attn_fwd_kernel)
for n in 0 1 2 3 4 5 6 7 8; do
  if [ $n -eq 0 ]; then L=peneo_amd/lib/libpeneo_hip.so; else L=tools/_abl_$n.so; fi
  echo "== ablate $n"; PENEO_HIP_LIB=$L B=${B:-7} timeout 120 python tools/run_attn_nobias.py 2>&1 | grep "fwd"
done
