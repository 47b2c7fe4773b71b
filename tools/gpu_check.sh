cd /root/repo
timeout 900 python -m pytest tests/test_hip_network.py -m gpu -x -q -k "c5" 2>&1 | tail -15
