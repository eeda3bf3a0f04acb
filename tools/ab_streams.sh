python tools/run_ddp_world1.py > /dev/null 2>&1     # first touch of a fresh box is slow
for n in 3 2 1 3 2 1; do echo "SIDE_STREAMS=$n"; PENEO_SIDE_STREAMS=$n timeout 600 python tools/run_ddp_world1.py 2>&1 | grep "docs/s" | grep -v print; done
