#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06n
mkdir -p $OUT
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_nets_golden.py -q -m gpu 2>&1 | tail -5 > $OUT/tests.log
for z in 0 1; do
  LSI_IGEMM_SWZ=$z timeout 600 python tools/conv_bench.py --out $OUT/conv_bench_swz$z.json > $OUT/conv_bench_swz$z.txt 2>&1
done
for rep in 1 2; do for z in 0 1; do
  echo "swz=$z eager L4" >> $OUT/train_ab.txt
  LSI_IGEMM_SWZ=$z timeout 300 python tools/train_bench.py --bf16 true --n_layers 4 --steps 40 2>>$OUT/train_ab.err | tail -1 >> $OUT/train_ab.txt
done; done
tail -2 $OUT/tests.log; tail -1 $OUT/conv_bench_swz0.txt; tail -1 $OUT/conv_bench_swz1.txt; cut -c1-100 $OUT/train_ab.txt
