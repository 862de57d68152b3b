cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 300 python bench.py --workload cfg4 --no-extra --no-cpu-baseline --steps 50 --warmup 10 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline'])"
