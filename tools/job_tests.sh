cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tests
timeout 2700 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -25 > gpurun_out/tests/tests.txt
cat gpurun_out/tests/tests.txt
