cd /root/repo
timeout 1500 python -m pytest tests/test_hip_network.py -m gpu -x -q 2>&1 | tail -2
run() { echo "$@"; timeout 200 python bench.py --no_cpu_baseline "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'] if d.get('roofline') else '')"; }
run
run
run --diag_skip_final_fps --steps 6
