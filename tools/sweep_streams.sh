#!/bin/bash
# sweep of the level chunk size (TPU3_MAX_PATCHES) and the network streams of the bench (GPU box)
cd /root/repo
for mp in 4096 1024 512 256; do for ns in 8 4 2; do
  sb=$((32/ns)); [ $sb -gt 4 ] && sb=4
  echo "max_patches=$mp net_streams=$ns sub_batch=$sb: $(TPU3_MAX_PATCHES=$mp python bench.py --no_cpu_baseline --no_extras --steps 6 --warmup 1 --net_streams $ns --sub_batch $sb 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e6,3), round(d["ms_per_step"],1))')"
done; done
