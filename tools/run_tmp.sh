O=gpurun_out/r5; mkdir -p $O
(timeout 1500 python -m pytest tests/test_bench_multirank.py tests/test_hip_network.py tests/test_hip_kernels.py -q -x -k "eight or graphed or host_tensors or cluster_fps or fed_back" 2>&1 | tail -15) > $O/gputest_b.txt
timeout 900 python bench.py > $O/bench_b.json 2> $O/bench_b.err
cat $O/gputest_b.txt; tail -3 $O/bench_b.err; python - <<'PY'
import json
l=json.loads(open("gpurun_out/r5/bench_b.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","value_1cloud","ms_per_cloud_1cloud","value_8clouds"): print(k, l.get(k))
print(json.dumps(l.get("roofline"))[:900])
ex=l.get("extras",{})
print({k:ex.get(k) for k in ("latency_ms_1cloud","latency_ms_1cloud_eager","latency_1cloud_form","ms_per_step_8clouds","train_step_ms")})
for o in l.get("rooflines_other",[]):
    print(o["kernel"][:70], o.get("ms_per_step"), o.get("frac"), o.get("x_over_floor"))
PY
