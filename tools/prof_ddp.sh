cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ddp -o run -- python3 tools/run_ddp_world1.py > gpurun_out/prof_ddp.log 2>&1
python tools/prof_summary.py gpurun_out/prof_ddp/run_results.db 45 | grep -v "peneo::" | head -30
