mkdir -p gpurun_out/r5
for cfg in "32 8 4" "48 8 6" "64 8 8" "64 16 4" "96 8 12"; do
  set -- $cfg
  echo -n "clouds=$1 net_streams=$2 sub_batch=$3 : "
  timeout 400 python bench.py --no_cpu_baseline --no_extras --clouds $1 --net_streams $2 --sub_batch $3 --steps 10 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f ms/step  %.3f M pts/s' % (l['ms_per_step'], l['value']/1e6))"
done > gpurun_out/r5/sweep_clouds.txt 2>&1
cat gpurun_out/r5/sweep_clouds.txt
