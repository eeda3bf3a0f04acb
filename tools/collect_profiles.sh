#!/bin/bash
# Collects the round's judged numbers on the GPU box into gpurun_out/final/ (run through gpurun; copy into profiles/ afterwards).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final; mkdir -p $OUT
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1            # first touch of a fresh box is slow
python bench.py > $OUT/default_line.json 2> $OUT/default.err
rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $OUT/prof -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/steps5_line.json 2> $OUT/steps5.err
python bench.py --backbone lilt --no-cpu-baseline > $OUT/lilt_line.json 2>/dev/null
python bench.py --size large --seq-len 1024 --lines 256 --docs-per-gpu 2 --no-cpu-baseline > $OUT/large_line.json 2>/dev/null
python bench.py --dtype fp32 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/fp32_line.json 2>/dev/null
python tools/run_blas_ref.py > $OUT/gemm_vs_vendor.txt 2>&1
python tools/run_phases.py > $OUT/phases.txt 2>&1
python tools/run_cpu_bound.py > $OUT/cpu_bound.txt 2>&1
FULL=1 python tools/run_decoder_bwd.py > $OUT/decoder_bwd.txt 2>&1
ls -la $OUT $OUT/prof | head -40
