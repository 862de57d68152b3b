#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06u
mkdir -p $OUT
timeout 600 python tools/host_profile.py --n_layers 4 2>&1 | grep -v amdgpu | head -12 > $OUT/host_profile_L4.txt
for L in 4 2; do for rep in 1 2 3; do
  echo "L$L eager" >> $OUT/train.txt
  timeout 300 python tools/train_bench.py --bf16 true --n_layers $L --steps 40 2>/dev/null | tail -1 >> $OUT/train.txt
done; done
timeout 600 python -m pytest tests/test_train_gpu.py tests/test_splat_gpu.py -q -m gpu -x 2>&1 | tail -3 >> $OUT/train.txt
head -3 $OUT/host_profile_L4.txt; cut -c1-100 $OUT/train.txt
