#!/bin/bash
# development sweep: final-FPS side streams / hardware queues
for fs in 1 2 3 4; do
  python bench.py --steps 8 --warmup 3 --no_cpu_baseline --fps_streams $fs 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams $fs', round(d['value']), round(d['ms_per_step'],1), round(d['roofline']['launch_ms'],1))"
done
GPU_MAX_HW_QUEUES=8 python bench.py --steps 8 --warmup 3 --no_cpu_baseline --fps_streams 4 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hwq8 streams 4', round(d['value']), round(d['ms_per_step'],1), round(d['roofline']['launch_ms'],1))"
python bench.py --steps 8 --warmup 3 --no_cpu_baseline --fps_streams 4 --clouds 16 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('clouds16 streams 4', round(d['value']), round(d['ms_per_step'],1), round(d['roofline']['launch_ms'],1))"
