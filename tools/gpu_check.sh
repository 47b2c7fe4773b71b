cd /root/repo
python tools/knn_norm_prune_probe.py 2>&1 | tail -20
