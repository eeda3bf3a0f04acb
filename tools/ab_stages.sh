for s in 3 4 2; do echo "GEMM_STAGES=$s"; PENEO_GEMM_STAGES=$s python tools/run_blas_ref.py 2>&1 | grep -E "peneo" | head -3; done
