cd /root/repo
for cfg in "--clouds 32" "--clouds 64 --net_streams 8 --sub_batch 4" "--clouds 64 --net_streams 16 --sub_batch 4" "--clouds 48 --net_streams 12 --sub_batch 4"; do
  echo -n "$cfg: "; timeout 200 python bench.py --no_cpu_baseline --steps 12 $cfg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f M  %.1f ms/step  fps launch %.0f ms' % (d['value']/1e6, d['ms_per_step'], d['roofline']['launch_ms']))" || echo failed
done
