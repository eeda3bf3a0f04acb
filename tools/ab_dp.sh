for c in 1 2 1 2; do echo "DP_CHUNKS=$c"; PENEO_DP_CHUNKS=$c timeout 600 python tools/run_ddp_world1.py 2>&1 | grep "docs/s" | grep -v print | head -2; done
