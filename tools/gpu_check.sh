cd /root/repo
timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_hip_network.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --no_cpu_baseline --steps 8 --warmup 2 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"
timeout 300 python bench.py --no_cpu_baseline --diag_skip_final_fps --steps 6 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
