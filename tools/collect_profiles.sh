#!/bin/bash
# Round-end evidence run (GPU box): default bench, rocprofv3 kernel stats of the same command,
# single-stream kernel stats, the FETCH_SIZE / WRITE_SIZE PMC passes and the SQ (MFMA / VALU busy) passes.
# Every PMC pass is its own run with --kernel-trace only.  Output: gpurun_out/final/
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/final
rm -rf $O gpurun_out/prof; mkdir -p $O gpurun_out/prof
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof/a -- python /root/repo/bench.py --no_cpu_baseline --no_extras > /root/repo/$O/bench_default_under_rocprof.json 2>/dev/null)
cp $(find gpurun_out/prof/a -name '*kernel_stats.csv' | head -1) $O/kernel_stats_bench_default.csv
SS="--no_cpu_baseline --no_extras --no_overlap --net_streams 1"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof/b -- python /root/repo/bench.py $SS --steps 4 --warmup 1 > /root/repo/$O/bench_single_stream_under_rocprof.json 2>/dev/null)
cp $(find gpurun_out/prof/b -name '*kernel_stats.csv' | head -1) $O/kernel_stats_single_stream.csv
pass() { # $1 = tag, rest = counters
  tag=$1; shift
  (cd /tmp && timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /root/repo/gpurun_out/prof/$tag -- python /root/repo/bench.py $SS --steps 2 --warmup 1 > /dev/null 2>&1)
  f=$(find gpurun_out/prof/$tag -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_by_kernel.py $f > $O/pmc_${tag}_by_kernel.txt
  [ -n "$f" ] && { head -1 $f > $O/pmc_${tag}_fm_main.csv; grep -E 'fl_main_kernel|fm_main_kernel' $f >> $O/pmc_${tag}_fm_main.csv; }
  rm -rf gpurun_out/prof/$tag
}
pass FETCH_SIZE FETCH_SIZE
pass WRITE_SIZE WRITE_SIZE
pass SQ_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA
pass SQ_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_ANY
pass GRBM GRBM_GUI_ACTIVE GRBM_COUNT
# (r6) the point-set kernels outside the step -- nm-distance fwd (scan and grid form) / bwd, ball query, gather: kernel
# stats and three PMC passes of tools/losses_probe.py
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof/l -- python /root/repo/tools/losses_probe.py > /dev/null 2>&1)
cp $(find gpurun_out/prof/l -name '*kernel_stats.csv' | head -1) $O/kernel_stats_losses.csv
lpass() { # $1 = tag, rest = counters
  tag=$1; shift
  (cd /tmp && timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /root/repo/gpurun_out/prof/l$tag -- python /root/repo/tools/losses_probe.py --reps 3 > /dev/null 2>&1)
  f=$(find gpurun_out/prof/l$tag -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_by_kernel.py $f > $O/pmc_${tag}_losses_by_kernel.txt
  rm -rf gpurun_out/prof/l$tag
}
lpass SQ_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU
lpass FETCH_SIZE FETCH_SIZE
lpass WRITE_SIZE WRITE_SIZE
tail -1 $O/bench_default.json | cut -c1-300
python tools/kstats.py $O/kernel_stats_single_stream.csv 5 16
python tools/kstats.py $O/kernel_stats_losses.csv 20 30
grep -h "fl_main\|fm_main\|dec_fused\|regress_tail\|linear_small\|linear_wide\|wide_split\|knn_graph\|knn_slab\|rl_main\|skip_" $O/pmc_*_by_kernel.txt | cut -c1-260
rm -rf gpurun_out/prof
