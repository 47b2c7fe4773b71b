mkdir -p gpurun_out/r5
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_network.py tests/test_c2_parity.py -q -x 2>&1 | tail -4) > gpurun_out/r5/tiles_sorted.txt
timeout 600 python tools/knn_tiles_probe.py >> gpurun_out/r5/tiles_sorted.txt 2>&1
for env in "X=0" "TPU3_KNN_TILES_SORT=0" "TPU3_KNN_TILES_MIN_N=16384"; do
  echo -n "$env : "
  env $env timeout 300 python bench.py --no_cpu_baseline --no_extras 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f ms/step  %.3f M pts/s' % (l['ms_per_step'], l['value']/1e6))"
done >> gpurun_out/r5/tiles_sorted.txt 2>&1
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof/t -- python /root/repo/bench.py --no_cpu_baseline --no_extras --no_overlap --net_streams 1 --steps 4 --warmup 1 > /dev/null 2>&1)
python tools/kstats.py $(find gpurun_out/prof/t -name '*kernel_stats.csv' | head -1) 5 40 | grep -i "kt_\|knn_insert\|knn_dup\|knn_compact\|total" >> gpurun_out/r5/tiles_sorted.txt; rm -rf gpurun_out/prof
cat gpurun_out/r5/tiles_sorted.txt | cut -c1-200
