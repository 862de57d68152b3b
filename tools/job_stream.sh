# usage: bash tools/job_stream.sh <tag>   -- stream-path parity subset + benches
cd $GRAFT_REPO_ROOT
TAG=${1:-x}
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests/test_splat_gpu.py tests/test_full_size_gpu.py -x -q -k "stream or kitti or cfg5 or golden or full_size or config2" > gpurun_out/$TAG/tests.log 2>&1
tail -12 gpurun_out/$TAG/tests.log
for wl in cfg3 cfg2 cfg5; do
  LSI_STREAM_VERBOSE=1 timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-extra --steps 100 > gpurun_out/$TAG/bench_$wl.json 2> gpurun_out/$TAG/bench_$wl.err
  python - <<PY
import json
try:
    r=json.load(open("gpurun_out/$TAG/bench_$wl.json"))
    print("$wl", round(r["roofline"]["avg_launch_us"],2), "us frac", round(r["roofline"]["frac"],4))
except Exception as e:
    print("$wl failed", e)
PY
  grep "plan" gpurun_out/$TAG/bench_$wl.err | head -1
done
