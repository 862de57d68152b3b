#!/bin/bash
# round 6: BN second passes from the tensor's end (A/B), the split convolutions in
# the training step (A/B), the headline's two-launch / alignment experiments
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06d
mkdir -p $OUT
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_train_gpu.py tests/test_nets_golden.py -q -m gpu -n 4 2>&1 | tail -60 > $OUT/tests.log
LSI_BN_REVERSE=1 timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_train_gpu.py -q -m gpu -n 4 2>&1 | tail -30 > $OUT/tests_rev.log
for rep in 1 2; do
for cfg in "0 512" "1 512" "0 0" "1 0"; do
  set -- $cfg
  for g in true false; do
  echo "rev=$1 splitk=$2 graph=$g" >> $OUT/train_ab.txt
  LSI_BN_REVERSE=$1 LSI_IGEMM_SPLITK=$2 timeout 300 python tools/train_bench.py --bf16 true --n_layers 4 --steps 40 --hip_graph $g 2>/dev/null | tail -1 >> $OUT/train_ab.txt
  done
done
done
timeout 600 python tools/headline_two_streams.py > $OUT/headline_two_streams.json 2> $OUT/headline_two_streams.err
tail -3 $OUT/tests.log; tail -3 $OUT/tests_rev.log; cat $OUT/train_ab.txt | cut -c1-120; cat $OUT/headline_two_streams.json; tail -5 $OUT/headline_two_streams.err
