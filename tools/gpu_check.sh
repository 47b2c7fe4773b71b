cd /root/repo
run() { echo "$@"; timeout 300 python bench.py --no_cpu_baseline "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'] if d.get('roofline') else '')"; }
run --diag_skip_final_fps
run --steps 30
run --net_streams 8 --sub_batch 2
run --net_streams 2 --sub_batch 8
run --fps_streams 8
