cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 100 python bench.py --workload cfg3 --no-extra --no-cpu-baseline --steps 200 --warmup 20 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('cfg3', round(j['ms_per_step']*1000,1))"
done
