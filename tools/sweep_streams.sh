cd /root/repo
for cfg in "--steps 20" "--steps 20 --fps_per_sub_batch --fps_streams 8" "--steps 5" "--steps 5 --fps_per_sub_batch --fps_streams 8" "--steps 20 --fps_per_sub_batch --fps_streams 16"; do
  echo -n "$cfg: "; timeout 150 python bench.py --no_cpu_baseline $cfg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f M  %.1f ms/step  fps launch %.0f ms' % (d['value']/1e6, d['ms_per_step'], d['roofline']['launch_ms']))" || echo failed
done
