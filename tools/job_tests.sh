cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03t
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r03t/tests.txt
cat gpurun_out/r03t/tests.txt
