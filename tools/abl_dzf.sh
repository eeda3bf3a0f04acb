for a in 0 1 2 4 8 3 7 15; do echo "ablate $a"; PENEO_DZF_ABLATE=$a FULL=1 timeout 300 python tools/run_decoder_bwd.py 2>&1 | grep -E "fused whole"; done
