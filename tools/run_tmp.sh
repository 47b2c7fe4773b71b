mkdir -p gpurun_out/r5
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_hip_kernels.py -q -k "knn_graph" 2>&1 | tail -3; CHECK=1 timeout 300 python tools/knn_slab_probe.py) > gpurun_out/r5/slab_probe3.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof/k -- python /root/repo/tools/knn_slab_probe.py > /dev/null 2>&1)
python tools/kstats.py $(find gpurun_out/prof/k -name '*kernel_stats.csv' | head -1) 1 4 >> gpurun_out/r5/slab_probe3.txt; rm -rf gpurun_out/prof
cat gpurun_out/r5/slab_probe3.txt | cut -c1-170
