cd /root/repo
timeout 600 python tools/c5_probe.py 2>&1 | tail -3
