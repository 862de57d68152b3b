#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06o
mkdir -p $OUT
python tools/graph_branch_probe.py 50 2>/dev/null > $OUT/graph_branch_probe.txt
python tools/graph_branch_probe.py 200 2>/dev/null >> $OUT/graph_branch_probe.txt
timeout 900 python -m pytest tests/test_train_gpu.py -q -m gpu -k "full_size_own_kernels or config_4_shape" 2>&1 | tail -15 > $OUT/tests.log
cat $OUT/graph_branch_probe.txt; tail -5 $OUT/tests.log
