cd /root/repo
timeout 1500 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -2
timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
