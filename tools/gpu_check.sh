cd /root/repo
timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_hip_network.py -m gpu -x -q -k "knn or graph or net or level or dec or edge or c5" 2>&1 | tail -2
python tools/knn_graph_probe.py
run() { echo "$@"; timeout 300 python bench.py --no_cpu_baseline "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run
