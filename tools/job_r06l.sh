#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06l
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python $R/tools/train_bench.py --bf16 true --n_layers 4 --steps 8 > $OUT/train.json 2> $OUT/kt.err
find $OUT/kt -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/kt
tail -1 $OUT/train.json
