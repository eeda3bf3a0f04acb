#!/bin/bash
# interleaved A/B of the working tree against a checkout of an older commit in ./_old (git worktree add _old <rev>; make there)
for i in 1 2 3; do
  for d in . _old; do
    ( cd $d && python bench.py --no-cpu-baseline --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d', d['value'], d['ms_per_step'], d.get('forward_only',{}).get('ms'))" )
  done
done
