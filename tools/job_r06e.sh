#!/bin/bash
# round 6: weight gradients on a side stream (A/B), the register-stencil disparity
# regulariser backward, full GPU suite
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06e
mkdir -p $OUT
( time timeout 1500 python -m pytest tests/ -q -m gpu -n 4 2>&1 | tail -80 ) > $OUT/gputest.log 2>&1
for rep in 1 2; do
for ws in 1 0; do
  for g in true false; do
  echo "wgrad_stream=$ws graph=$g" >> $OUT/train_ab.txt
  LSI_WGRAD_STREAM=$ws timeout 300 python tools/train_bench.py --bf16 true --n_layers 4 --steps 40 --hip_graph $g 2>>$OUT/train_ab.err | tail -1 >> $OUT/train_ab.txt
  done
done
done
for ws in 1 0; do
  echo "L2 wgrad_stream=$ws graph=true" >> $OUT/train_ab.txt
  LSI_WGRAD_STREAM=$ws timeout 300 python tools/train_bench.py --bf16 true --n_layers 2 --steps 40 --hip_graph true 2>>$OUT/train_ab.err | tail -1 >> $OUT/train_ab.txt
done
tail -6 $OUT/gputest.log; cat $OUT/train_ab.txt | cut -c1-100; tail -5 $OUT/train_ab.err
