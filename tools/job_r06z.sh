#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06z
mkdir -p $OUT
for rep in 1 2; do for gp in unet 0 heads; do
  echo "graph_parts=$gp L4 eager" >> $OUT/train_ab.txt
  LSI_GRAPH_PARTS=$gp timeout 400 python tools/train_bench.py --bf16 true --n_layers 4 --steps 40 2>>$OUT/err.txt | tail -1 >> $OUT/train_ab.txt
done; done
for gp in unet 0; do
  echo "graph_parts=$gp L2 eager" >> $OUT/train_ab.txt
  LSI_GRAPH_PARTS=$gp timeout 400 python tools/train_bench.py --bf16 true --n_layers 2 --steps 40 2>>$OUT/err.txt | tail -1 >> $OUT/train_ab.txt
done
cut -c1-100 $OUT/train_ab.txt
