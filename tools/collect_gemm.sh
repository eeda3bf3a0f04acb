#!/bin/bash
# GEMM evidence of round 3: vendor-library comparison on the encoder / decoder shapes, tile-shape table, PMC of the two kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/gemm; rm -rf $OUT; mkdir -p $OUT
python tools/run_blas_ref.py > $OUT/gemm_vs_vendor_library.txt 2>&1
python tools/run_gemm_big.py > $OUT/gemm_tile_shapes.txt 2>&1
for s in ffn1 qkv ffn2; do
  echo "== $s: gemm_big (auto)" >> $OUT/gemm_pmc.txt
  SHAPE=$s bash tools/pmc.sh gemm tools/run_gemm_once.py >> $OUT/gemm_pmc.txt 2>&1
  echo "== $s: 128 x 128 kernel (PENEO_GEMM_BIG=0)" >> $OUT/gemm_pmc.txt
  SHAPE=$s PENEO_GEMM_BIG=0 bash tools/pmc.sh gemm_dma_pipe tools/run_gemm_once.py >> $OUT/gemm_pmc.txt 2>&1
done
tail -30 $OUT/gemm_vs_vendor_library.txt
