#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06r
mkdir -p $OUT
for rep in 1 2 3; do for z in 0 1 2; do
  echo "wgrad_swz=$z eager L4" >> $OUT/train_ab.txt
  LSI_WGRAD_SWZ=$z timeout 300 python tools/train_bench.py --bf16 true --n_layers 4 --steps 40 2>>$OUT/train_ab.err | tail -1 >> $OUT/train_ab.txt
done; done
cut -c1-100 $OUT/train_ab.txt
