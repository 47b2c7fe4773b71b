#!/bin/bash
# Round-end evidence run (GPU box): default bench, rocprofv3 kernel stats of the same command,
# single-stream kernel stats, and the FETCH_SIZE / WRITE_SIZE PMC passes.  Output: gpurun_out/final/
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/final
rm -rf $O gpurun_out/prof; mkdir -p $O gpurun_out/prof
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof/a -- python /root/repo/bench.py --no_cpu_baseline > /root/repo/$O/bench_default_under_rocprof.json 2>/dev/null)
cp $(find gpurun_out/prof/a -name '*kernel_stats.csv' | head -1) $O/kernel_stats_bench_default.csv
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof/b -- python /root/repo/bench.py --no_cpu_baseline --no_overlap --net_streams 1 --steps 4 --warmup 1 > /root/repo/$O/bench_single_stream_under_rocprof.json 2>/dev/null)
cp $(find gpurun_out/prof/b -name '*kernel_stats.csv' | head -1) $O/kernel_stats_single_stream.csv
for c in FETCH_SIZE WRITE_SIZE; do # (PMC passes: 2 steps are enough)
  (cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /root/repo/gpurun_out/prof/$c -- python /root/repo/bench.py --no_cpu_baseline --no_overlap --net_streams 1 --steps 2 --warmup 1 > /dev/null 2>&1)
  f=$(find gpurun_out/prof/$c -name '*counter_collection.csv' | head -1)
  python tools/pmc_by_kernel.py $f > $O/pmc_${c}_by_kernel.txt
  head -1 $f > $O/pmc_${c}_fb_main.csv; grep fb_main_kernel $f >> $O/pmc_${c}_fb_main.csv
done
tail -1 $O/bench_default.json | cut -c1-400
python tools/kstats.py $O/kernel_stats_single_stream.csv 5 14
grep fb_main $O/pmc_FETCH_SIZE_by_kernel.txt $O/pmc_WRITE_SIZE_by_kernel.txt
rm -rf gpurun_out/prof gpurun_out/pmc
