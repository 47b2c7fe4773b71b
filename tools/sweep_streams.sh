cd /root/repo
for cfg in "--net_streams 8 --sub_batch 4" "--net_streams 8 --sub_batch 2" "--net_streams 16 --sub_batch 2" "--net_streams 4 --sub_batch 4" "--clouds 48 --net_streams 12 --sub_batch 4" "--fps_streams 2"; do
  echo -n "$cfg: "; timeout 120 python bench.py --no_cpu_baseline --steps 10 $cfg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f M  %.1f ms/step  fps launch %.0f ms' % (d['value']/1e6, d['ms_per_step'], d['roofline']['launch_ms']))" || echo failed
done
