cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_splat_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -k "tile or any_pose or hostile" 2>&1 | tail -2
for f in 0 4096 0 4096; do
timeout 200 python bench.py --workload cfg4 --no-extra --no-cpu-baseline --steps 100 --warmup 10 --debug-flags $f 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('cfg4 flags $f', round(j['ms_per_step']*1000,1))"
done
timeout 200 python tools/sweep_probe.py cfg4 0 2>&1 | grep -E "per-WG|within"
