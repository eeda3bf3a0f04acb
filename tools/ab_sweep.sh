#!/bin/bash
# Interleaved sweep of ONE environment switch on one box (run through gpurun), the parametrised form of the round 1-3 one-off
# A/B scripts:   bash tools/ab_sweep.sh VAR v1 v2 ... [-- command ...]
#   bash tools/ab_sweep.sh PENEO_STAGE_CALLS 0 1                                  # bench.py: docs/s, ms/step, eval-forward ms
#   bash tools/ab_sweep.sh PENEO_DW1_SPLIT 0 3 5 -- python bench.py --docs-per-gpu 12 --no-cpu-baseline
#   bash tools/ab_sweep.sh B 7 8 14 -- python tools/run_attn.py                   # any script that reads the variable
# ROUNDS=n (default 2) interleaved rounds.  Two builds of the tree: tools/ab_old_new.sh; two builds of the library: ab_lib.sh.
VAR=$1; shift
VALS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done
[ "$1" == "--" ] && shift
cd $GRAFT_REPO_ROOT
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ragged > /dev/null 2>&1      # first touch of a fresh box is slow
for round in $(seq ${ROUNDS:-2}); do
  for v in "${VALS[@]}"; do
    echo -n "$VAR=$v  "
    if [ $# -gt 0 ]; then env $VAR=$v "$@" 2>&1 | grep -v amdgpu.ids | tail -${TAIL:-1}
    else env $VAR=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ragged 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['forward_only']['ms_per_batch'])"
    fi
  done
done
