mkdir -p gpurun_out/r5
(timeout 600 python -m pytest tests/test_hip_kernels.py -q -k "knn_graph" 2>&1 | tail -3; CHECK=1 timeout 300 python tools/knn_slab_probe.py) > gpurun_out/r5/slab_probe2.txt 2>&1; cat gpurun_out/r5/slab_probe2.txt
