mkdir -p gpurun_out/r5
for env in "X=0" "TPU3_DEC_PERSIST=3" "TPU3_DEC_PERSIST=2" "TPU3_DEC_PERSIST=6"; do
  echo -n "$env : "
  env $env timeout 300 python bench.py --no_cpu_baseline --no_extras 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f ms/step  %.3f M pts/s' % (l['ms_per_step'], l['value']/1e6))"
done > gpurun_out/r5/dec_hooks.txt 2>&1
(timeout 600 python -m pytest tests/test_hip_network.py -q -x -k "regress or level_forward or net_eval or teacher" 2>&1 | tail -3) >> gpurun_out/r5/dec_hooks.txt
cat gpurun_out/r5/dec_hooks.txt
