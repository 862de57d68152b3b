#!/bin/bash
# round 6 profile collection: splat side (collect_profiles.sh), then the training step
cd $GRAFT_REPO_ROOT
bash tools/build_variant.sh stamps lsi_splat_stream2.hip -DS2X_STAMPS > gpurun_out/build_stamps.log 2>&1
bash tools/collect_profiles.sh r06 > gpurun_out/collect_r06.log 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06
for L in 4 2; do
  for g in false true; do
    for rep in 1 2 3; do
      timeout 300 python $R/tools/train_bench.py --bf16 true --n_layers $L --steps 40 --hip_graph $g 2>/dev/null | tail -1 >> $OUT/train_bf16_L${L}_graph_$g.jsonl
    done
  done
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_train -o kt -- python $R/tools/train_bench.py --bf16 true --n_layers 4 --steps 15 --hip_graph true > $OUT/train_L4_graph_profiled.json 2> $OUT/kt_train.err
find $OUT/kt_train -name "*kernel_stats.csv" -exec cp {} $OUT/train_L4_kernel_stats.csv \;
rm -rf $OUT/kt_train
timeout 600 python $R/tools/conv_bench.py --out $OUT/conv_bench.json > $OUT/conv_bench.txt 2>&1
timeout 600 python $R/tools/conv_util.py --bf16 true --n_layers 2 > $OUT/conv_util_bf16.json 2> $OUT/conv_util.err
tail -3 $OUT/conv_bench.txt; cat $OUT/train_bf16_L4_graph_false.jsonl | cut -c1-90
