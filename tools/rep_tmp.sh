run() { env "$@" PENEO_BENCH_ALLOC_STATS=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ragged 2>&1 | grep -v amdgpu | python -c "
import sys, json
h = ''
for l in sys.stdin:
    if l.startswith('alloc'): h = l[l.index('host ms'):].strip()[:110]
    elif l.startswith('dW1'): w = l[l.index('['):].strip()[:60]
    elif l.startswith('{'): d = json.loads(l); print('$*', d['value'], d['ms_per_step'], 'dW1', w, h)
"; }
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ragged > /dev/null 2>&1
for r in 1 2 3; do
  run PENEO_BENCH_PRIORITY=1 PENEO_PAIR_SAVE=1
  run PENEO_BENCH_PRIORITY=0 PENEO_PAIR_SAVE=1
  run PENEO_BENCH_PRIORITY=1 PENEO_PAIR_SAVE=0
done
