cd /root/repo
run() { echo "nogs=$TPU3_FPS_NO_GLOBAL_SORT $@"; timeout 300 python bench.py --no_cpu_baseline "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'] if d.get('roofline') else '', d['roofline'].get('operator_ms'))"; }
run; export TPU3_FPS_NO_GLOBAL_SORT=1; run; unset TPU3_FPS_NO_GLOBAL_SORT; run; export TPU3_FPS_NO_GLOBAL_SORT=1; run
run --steps 5 --warmup 1; unset TPU3_FPS_NO_GLOBAL_SORT; run --steps 5 --warmup 1
