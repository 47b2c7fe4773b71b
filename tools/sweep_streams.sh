#!/bin/bash
# sweep of the bench's clouds per step / network streams / clouds per sub-batch (GPU box)
cd /root/repo
for cfg in "32 8 4" "40 8 5" "35 7 5" "30 6 5" "40 10 4" "48 8 6"; do set -- $cfg
  echo "clouds=$1 net_streams=$2 sub_batch=$3: $(python bench.py --no_cpu_baseline --no_extras --clouds $1 --net_streams $2 --sub_batch $3 --steps 12 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e6,3), round(d["ms_per_step"],1))')"
done
