cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_sampling_gpu.py tests/test_abi.py -x -q 2>&1 | tail -4
