#!/bin/bash
# the default bench line and its rocprofv3 kernel stats, back to back on one box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06s
mkdir -p $OUT
timeout 900 python $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2>> $OUT/bench_default.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-extra --traffic off > $OUT/kt_bench.json 2> $OUT/kt.err
cp $OUT/kt/kt_kernel_stats.csv $OUT/kernel_stats_default.csv; rm -rf $OUT/kt
grep splat_stream2 $OUT/kernel_stats_default.csv | cut -c1-200
python3 -c "
import json
for f in ('bench_default.json','bench_driver_style.json','kt_bench.json'):
    d=json.load(open('$OUT/'+f)); print(f, d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
"
