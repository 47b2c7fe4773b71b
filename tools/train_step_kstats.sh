#!/bin/bash
# per-kernel totals of the bench's training step (tools/train_step_probe.py, eager so that every kernel is traced) (GPU box)
cd /root/repo
export TMPDIR=/tmp
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
(cd /tmp && RATIOS=${RATIOS:-16} MODES=eager STEPS=10 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof/t -- python /root/repo/tools/train_step_probe.py 2>/dev/null | tail -2)
python tools/kstats.py $(find gpurun_out/prof/t -name "*kernel_stats.csv" | head -1) 13 ${TOP:-60} | cut -c${CUT:-1-150}
rm -rf gpurun_out/prof
