#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06j
mkdir -p $OUT
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_nets_golden.py -q -m gpu 2>&1 | tail -8 > $OUT/tests.log
for rep in 1 2; do
for hs in 1 0; do
  echo "head_streams=$hs eager L4" >> $OUT/train_ab.txt
  LSI_HEAD_STREAMS=$hs timeout 300 python tools/train_bench.py --bf16 true --n_layers 4 --steps 40 2>>$OUT/train_ab.err | tail -1 >> $OUT/train_ab.txt
done
done
for hs in 1 0; do
  echo "head_streams=$hs eager L2" >> $OUT/train_ab.txt
  LSI_HEAD_STREAMS=$hs timeout 300 python tools/train_bench.py --bf16 true --n_layers 2 --steps 40 2>>$OUT/train_ab.err | tail -1 >> $OUT/train_ab.txt
done
tail -3 $OUT/tests.log; cat $OUT/train_ab.txt | cut -c1-100; tail -3 $OUT/train_ab.err
