#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06q
mkdir -p $OUT
timeout 900 python -m pytest tests/test_conv_gpu.py -q -m gpu 2>&1 | tail -5 > $OUT/tests.log
for z in 0 1; do
  LSI_WGRAD_SWZ=$z timeout 600 python tools/conv_bench.py --out $OUT/conv_bench_wswz$z.json > $OUT/conv_bench_wswz$z.txt 2>&1
done
for rep in 1 2; do for z in 0 1; do
  echo "wgrad_swz=$z eager L4" >> $OUT/train_ab.txt
  LSI_WGRAD_SWZ=$z timeout 300 python tools/train_bench.py --bf16 true --n_layers 4 --steps 40 2>>$OUT/train_ab.err | tail -1 >> $OUT/train_ab.txt
done; done
tail -2 $OUT/tests.log; tail -1 $OUT/conv_bench_wswz0.txt; tail -1 $OUT/conv_bench_wswz1.txt; cut -c1-100 $OUT/train_ab.txt
