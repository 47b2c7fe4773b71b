cd /root/repo
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k fps 2>&1 | tail -2
python tools/fps_probe.py 2>&1 | head -8
