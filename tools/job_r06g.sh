#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06g
mkdir -p $OUT
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_conv_gpu.py -q -m gpu -k "batch_norm or bn or statistics" 2>&1 | tail -8 > $OUT/tests.log
for rep in 1 2; do
for ms in 16 4 2 1; do
  echo "bn_min_steps=$ms eager" >> $OUT/train_ab.txt
  LSI_BN_MIN_STEPS=$ms timeout 300 python tools/train_bench.py --bf16 true --n_layers 4 --steps 40 2>>$OUT/train_ab.err | tail -1 >> $OUT/train_ab.txt
done
done
for ms in 16 2; do
  echo "bn_min_steps=$ms graph" >> $OUT/train_ab.txt
  LSI_BN_MIN_STEPS=$ms timeout 300 python tools/train_bench.py --bf16 true --n_layers 4 --steps 40 --hip_graph true 2>>$OUT/train_ab.err | tail -1 >> $OUT/train_ab.txt
done
tail -3 $OUT/tests.log; cat $OUT/train_ab.txt | cut -c1-100
