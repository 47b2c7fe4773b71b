cd /root/repo
timeout 600 python tools/chamfer_probe.py 2>&1 | tail -4
