cd /root/repo
timeout 1200 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "knn" 2>&1 | tail -2
run() { echo "$@"; timeout 300 python bench.py --no_cpu_baseline "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run
