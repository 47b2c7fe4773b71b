cd /root/repo
run() { echo "HWQ=$GPU_MAX_HW_QUEUES $@"; timeout 300 python bench.py --no_cpu_baseline --steps 12 --warmup 2 "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'] if d.get('roofline') else '')"; }
export GPU_MAX_HW_QUEUES=8
run --net_streams 2 --fps_streams 4
run --net_streams 3 --fps_streams 2
run --net_streams 4 --fps_streams 2
run --net_streams 1 --fps_streams 4
export GPU_MAX_HW_QUEUES=16
run --net_streams 2 --fps_streams 4
run --net_streams 4 --fps_streams 4
export GPU_MAX_HW_QUEUES=24
run --net_streams 2 --fps_streams 4
