#!/bin/bash
# round 6, first GPU job: the whole GPU suite, the per-stage bf16 errors, the
# default bench line (with extra.train_step) and the training step's kernel stats
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06a
mkdir -p $OUT
( time timeout 1500 python -m pytest tests/ -q -m gpu --maxfail=25 2>&1 | tail -60 ) > $OUT/gputest.log 2>&1
timeout 300 python tools/nets_stage_errors.py > $OUT/nets_stage_errors.txt 2> $OUT/nets_stage_errors.err
( time timeout 900 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt_train -o kt -- python $R/tools/train_bench.py --bf16 true --n_layers 4 --steps 10 > $R/$OUT/train_L4_eager.json 2> $R/$OUT/kt_train.err
cd $R
find $OUT/kt_train -name "*kernel_stats.csv" -exec cp {} $OUT/train_L4_kernel_stats.csv \;
rm -rf $OUT/kt_train
tail -5 $OUT/gputest.log; cat $OUT/nets_stage_errors.txt | head -70; tail -3 $OUT/bench_default.err; cat $OUT/bench_default.json | cut -c1-600
