# per-kernel durations with the decoder backward strictly serial (no overlap inflation); run through gpurun
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PENEO_DEC_STREAMS=1 PENEO_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_serial -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_serial.log 2>&1
python tools/prof_summary.py gpurun_out/prof_serial/run_results.db 30
