cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "fused_batch_norm_relu" 2>&1 | grep -E "^E|assert|passed|failed" | head -30
