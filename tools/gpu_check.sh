cd /root/repo
export TPU3_BENCH_BACKEND=gloo TPU3_BENCH_ONE_DEVICE=1
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --clouds 4 --no_cpu_baseline 2>&1 | tail -5 | cut -c1-600
