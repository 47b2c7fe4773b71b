cd /root/repo
timeout 1500 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --no_cpu_baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], [e['traffic'] for e in d['rooflines_other']])"
