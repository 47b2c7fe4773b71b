O=gpurun_out/r5; mkdir -p $O
(timeout 600 python -m pytest tests/test_hip_kernels.py -q -k "knn_graph" 2>&1 | tail -3) > $O/slab_test.txt
(for d in 0 7; do echo "DBG=$d random rows"; TPU3_KG_SLAB_DBG=$d ITERS=20 timeout 200 python tools/knn_graph_probe.py; done
 TPU3_KG_SLAB_DBG=0 CHECK=1 timeout 300 python tools/knn_slab_probe.py ) > $O/slab_probe.txt 2>&1
( timeout 300 python tools/fps_level_dispatch_probe.py
  TPU3_FPS_FORCE_TILE=1 TPU3_FPS_CLUSTER=0 timeout 600 python tools/fps_level_dispatch_probe.py
  TPU3_FPS_FORCE_TILE=1 TPU3_FPS_CLUSTER=2 SETS=48 timeout 300 python tools/fps_level_dispatch_probe.py
  TPU3_FPS_FORCE_TILE=1 TPU3_FPS_CLUSTER=4 SETS=48 timeout 300 python tools/fps_level_dispatch_probe.py ) > $O/fps_level_probe.txt 2>&1
cat $O/slab_test.txt $O/slab_probe.txt $O/fps_level_probe.txt
