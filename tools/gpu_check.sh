cd /root/repo
run() { echo "$@"; timeout 300 python bench.py --no_cpu_baseline "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'] if d.get('roofline') else '')"; }
run
run --clouds 1 --steps 6
timeout 300 python tools/c5_probe.py 2>&1 | tail -1
