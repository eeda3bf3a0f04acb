run() { echo "$@"; env "$@" PENEO_DIST_BACKEND=gloo PENEO_DEVICE=0 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c60-140; }
run PENEO_SIDE_STREAMS=3
run PENEO_SIDE_STREAMS=1
run PENEO_SIDE_STREAMS=2 GPU_MAX_HW_QUEUES=16
run PENEO_SIDE_STREAMS=3 PENEO_DP_IMPL=ddp
