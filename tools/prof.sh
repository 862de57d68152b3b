#!/bin/bash
# usage: tools/prof.sh <tag> <bench args...>   (run on the GPU box via gpurun)
# kernel-trace stats + two PMC passes (never combined with other trace domains).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $R/bench.py --no-cpu-baseline --steps 50 --launch eager "$@" > $OUT/bench_kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc1 -o p1 -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 --launch eager "$@" > $OUT/bench_p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS_ATOMIC --kernel-trace --output-format csv -d $OUT/pmc2 -o p2 -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 --launch eager "$@" > $OUT/bench_p2.log 2>&1
find $OUT -name "*.csv" | head -20
cat $OUT/kt/*kernel_stats.csv 2>/dev/null | head -8
python3 - <<PY
import csv, glob, collections
for p in ("pmc1","pmc2"):
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % p):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:40]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            n[(k,row["Counter_Name"])] += 1
        for k in agg:
            print(p, k, {c: round(v / n[(k,c)]) for c, v in agg[k].items()})
PY
