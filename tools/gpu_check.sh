cd /root/repo
echo resident-old; TPU3_FPS_BUCKET_MIN_N=100000 python tools/fps_real_probe.py | tail -3
