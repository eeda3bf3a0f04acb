#!/bin/bash
# Collects the round's judged numbers on the GPU box into gpurun_out/final/ (run through gpurun; copy into profiles/ afterwards).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final; rm -rf $OUT; mkdir -p $OUT
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1            # first touch of a fresh box is slow
python bench.py > $OUT/default_line.json 2> $OUT/default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/steps5_line.json 2> $OUT/steps5.err
T=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T +4 --gaps > $OUT/step_timeline.txt 2>&1
python tools/prof_summary_csv.py $T 40 > $OUT/steps5_summary.txt 2>&1
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/steps5_kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof
PENEO_PAIR_SAVE=0 python bench.py --no-cpu-baseline --no-ragged > $OUT/default_line_recompute.json 2> /dev/null    # round 5: the backward rebuilding z (the round-4 data flow)
(echo "-- pair_heads_fwd_hand_kernel<24, true, true> (saving form)"; WHICH=fwd bash tools/pmc_traffic.sh pair_heads_fwd_hand tools/run_pair_saved_once.py; echo "-- pair_bwd_sv_kernel<24, true>"; bash tools/pmc_traffic.sh pair_bwd_sv_kernel tools/run_pair_saved_once.py; python tools/run_pair_saved_once.py) > $OUT/pmc_pair_saved.txt 2>&1
TIME=1 python tools/check_pair_saved.py > $OUT/pair_saved_check.txt 2>&1
bash tools/pmc_traffic.sh pair_bwd_ws_kernel tools/run_pair_bwd_once.py > $OUT/pmc_pair_bwd.txt 2>&1
bash tools/pmc_traffic.sh pair_heads_fwd tools/run_pair.py > $OUT/pmc_pair_fwd.txt 2>&1
python tools/run_pair_bwd.py > $OUT/pair_bwd_kernel.txt 2>&1
bash tools/prof_hbm.sh final_hbm > /dev/null 2>&1; cp gpurun_out/final_hbm/hbm_kernels.txt $OUT/hbm_kernels.txt
python tools/run_host_time.py > $OUT/host_time.txt 2>&1
B=1 S=64 python tools/run_host_breakdown.py > $OUT/host_breakdown_small.txt 2>&1
python tools/run_blas_ref.py > $OUT/gemm_vs_vendor.txt 2>&1
python bench.py --backbone lilt --no-cpu-baseline > $OUT/lilt_line.json 2>/dev/null
python bench.py --vocab 250002 --no-cpu-baseline --no-ragged > $OUT/xlmr_vocab_line.json 2>/dev/null
python bench.py --size large --seq-len 1024 --lines 256 --docs-per-gpu 2 --no-cpu-baseline > $OUT/large_line.json 2>/dev/null
python bench.py --size large --seq-len 1024 --lines 256 --docs-per-gpu 4 --no-cpu-baseline --trained-agree-steps 0 > $OUT/large_4docs_line.json 2>/dev/null
python bench.py --dtype fp32 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/fp32_line.json 2>/dev/null
PENEO_DIST_BACKEND=gloo PENEO_DEVICE=0 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/two_ranks_one_gpu_gloo_line.json 2> $OUT/two_ranks.err
python tools/run_phases.py > $OUT/phases.txt 2>&1
ls -la $OUT | head -40
