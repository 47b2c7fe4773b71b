#!/bin/bash
# single-stream kernel statistics of the bench (GPU box): per-kernel ms per 32-cloud step
cd /root/repo
export TMPDIR=/tmp
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof/b -- python /root/repo/bench.py --no_cpu_baseline --no_extras --no_overlap --net_streams 1 --steps 4 --warmup 1 2>/dev/null | tail -1 | cut -c1-200)
python tools/kstats.py $(find gpurun_out/prof/b -name "*kernel_stats.csv" | head -1) 5 ${TOP:-22} | cut -c1-175
rm -rf gpurun_out/prof
