#!/bin/bash
# sweep of the bench's clouds per step / network streams / clouds per sub-batch / final-FPS side streams (GPU box)
cd /root/repo
for cfg in "32 8 4 4" "32 8 4 2" "32 8 4 8" "32 4 8 4" "32 16 2 4" "40 8 5 4" "48 8 6 4" "64 8 8 4" "64 16 4 4"; do set -- $cfg
  echo "clouds=$1 net_streams=$2 sub_batch=$3 fps_streams=$4: $(python bench.py --no_cpu_baseline --no_extras --clouds $1 --net_streams $2 --sub_batch $3 --fps_streams $4 --steps 12 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e6,3), "M points/s", round(d["ms_per_step"],1), "ms per step")')"
done
