#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06f
mkdir -p $OUT
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_loss_gpu.py tests/test_train_gpu.py -q -m gpu 2>&1 | tail -30 > $OUT/tests.log
timeout 900 python -m pytest tests/test_full_size_gpu.py tests/test_splat_gpu.py -q -m gpu -k "disparity_output_and_backward_full_size or differential_fuzz_of_the_round3" 2>&1 | tail -30 > $OUT/tests2.log
for rep in 1 2; do
for pr in 0 1; do
  echo "main_prio=$pr eager" >> $OUT/train_ab.txt
  LSI_MAIN_PRIO=$pr timeout 300 python tools/train_bench.py --bf16 true --n_layers 4 --steps 40 2>>$OUT/train_ab.err | tail -1 >> $OUT/train_ab.txt
done
done
tail -4 $OUT/tests.log; tail -4 $OUT/tests2.log; cat $OUT/train_ab.txt | cut -c1-100
