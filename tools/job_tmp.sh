cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_splat_gpu.py tests/test_train_gpu.py tests/test_abi.py -x -q -k "both or train or abi or atomic" 2>&1 | tail -15
