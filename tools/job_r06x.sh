#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06x
mkdir -p $OUT
for rep in 1 2; do for gp in 1 0; do
  echo "graph_parts=$gp L4 eager" >> $OUT/train_ab.txt
  LSI_GRAPH_PARTS=$gp timeout 400 python tools/train_bench.py --bf16 true --n_layers 4 --steps 40 2>>$OUT/err.txt | tail -1 >> $OUT/train_ab.txt
done; done
for gp in 1 0; do
  echo "graph_parts=$gp L2 eager" >> $OUT/train_ab.txt
  LSI_GRAPH_PARTS=$gp timeout 400 python tools/train_bench.py --bf16 true --n_layers 2 --steps 40 2>>$OUT/err.txt | tail -1 >> $OUT/train_ab.txt
done
timeout 900 python -m pytest tests/test_train_gpu.py -q -m gpu -x 2>&1 | tail -12 > $OUT/tests.log
cut -c1-100 $OUT/train_ab.txt; tail -5 $OUT/tests.log; grep -v amdgpu $OUT/err.txt | tail -12
