cd /root/repo
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_network.py -m gpu -x -q -k "nmdistance or chamfer" 2>&1 | tail -2
timeout 600 python tools/chamfer_probe.py 2>&1 | tail -4
