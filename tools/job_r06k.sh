#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06k
mkdir -p $OUT
for L in 4 2; do
for hs in 1 0; do
  echo "graph head_streams=$hs wgrad_stream=0 L$L" >> $OUT/train_ab.txt
  LSI_HEAD_STREAMS=$hs LSI_WGRAD_STREAM=0 timeout 300 python tools/train_bench.py --bf16 true --n_layers $L --steps 40 --hip_graph true 2>>$OUT/train_ab.err | tail -1 >> $OUT/train_ab.txt
done
done
cat $OUT/train_ab.txt | cut -c1-100; tail -3 $OUT/train_ab.err
