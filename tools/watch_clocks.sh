# Poll clocks / power while the training step runs (run through gpurun): bash tools/watch_clocks.sh
cd $GRAFT_REPO_ROOT
python bench.py --steps 400 --warmup 3 --no-cpu-baseline > /tmp/bench_long.json 2>/dev/null &
BP=$!
sleep 12
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (junction|edge)" | tr -s ' ' | head -8
  echo "--"
  sleep 0.7
done
wait $BP
python -c "import json; d=json.loads(open('/tmp/bench_long.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
echo "== idle"
sleep 2
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | head -4
