cd /root/repo
timeout 900 python -m pytest tests/test_hip_network.py -m gpu -x -q -k "wgrad or train" 2>&1 | tail -2
timeout 600 python tools/train_probe.py 2>&1 | tail -5
