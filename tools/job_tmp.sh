cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_splat_gpu.py -x -q -m gpu 2>&1 | tail -2
for wl in "cfg3 --shard-of 8" "cfg3 --shard-of 4" "cfg3 --shard-of 2" "cfg2" "cfg3" "cfg5"; do
LSI_STREAM_VERBOSE=1 timeout 100 python bench.py --workload $wl --no-extra --no-cpu-baseline --steps 100 --warmup 20 2>/tmp/err.txt | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$wl', round(j['ms_per_step']*1000,1))"
grep "stream plan" /tmp/err.txt | head -1
done
