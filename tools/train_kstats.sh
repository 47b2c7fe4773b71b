#!/bin/bash
# per-kernel totals of tools/train_probe.py (GPU box)
cd /root/repo
export TMPDIR=/tmp
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof/t -- python /root/repo/tools/train_probe.py 2>/dev/null | tail -6)
python tools/kstats.py $(find gpurun_out/prof/t -name "*kernel_stats.csv" | head -1) ${DIV:-1} 40 | cut -c1-160
rm -rf gpurun_out/prof
