for n in 128 256 512; do echo "LN_BWD_BLOCKS=$n"; PENEO_LN_BWD_BLOCKS=$n python tools/run_small_kernels.py 2>&1 | grep "ln_bwd"; done
python -m pytest tests/test_gpu_kernels.py -x -q -k "layernorm or ln" 2>&1 | tail -2
