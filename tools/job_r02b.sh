set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
timeout 900 python -m pytest tests/test_full_size_gpu.py -x -q > gpurun_out/r02b/fullsize.log 2>&1
tail -15 gpurun_out/r02b/fullsize.log
( time python bench.py ) > gpurun_out/r02b/bench_default.json 2> gpurun_out/r02b/bench_default.err
cat gpurun_out/r02b/bench_default.json; tail -5 gpurun_out/r02b/bench_default.err
