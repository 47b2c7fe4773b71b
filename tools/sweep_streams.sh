#!/bin/bash
# sweep of the bench's network streams / clouds per sub-batch (GPU box)
cd /root/repo
for cfg in "8 4" "16 2" "12 3" "8 2" "16 4" "4 8"; do set -- $cfg
  echo "net_streams=$1 sub_batch=$2: $(python bench.py --no_cpu_baseline --no_extras --net_streams $1 --sub_batch $2 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e6,3), round(d["ms_per_step"],1))')"
done
