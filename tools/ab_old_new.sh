#!/bin/bash
# interleaved A/B of the working tree against a checkout of an older commit in ./_old (git worktree add _old <rev>; make -C
# _old/peneo_amd/csrc): bash tools/ab_old_new.sh [rounds] [bench args]
R=${1:-3}; shift
for i in $(seq $R); do
  for d in . _old; do
    ( cd $d && python bench.py --no-cpu-baseline --no-ragged --steps 20 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d', d['value'], d['ms_per_step'], d['forward_only']['ms_per_batch'])" )
  done
done
