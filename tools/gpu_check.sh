cd /root/repo
run() { echo "$@"; timeout 150 python bench.py --no_cpu_baseline "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'] if d.get('roofline') else '')"; }
run
run --clouds 32 --steps 6
run --clouds 24 --steps 6
run --clouds 8
run --clouds 1 --steps 6
