#!/bin/bash
# usage: tools/pmc_mini.sh <kernel-substring> <bench args...>  -- instruction counts only
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
KN=$1; shift
rm -rf /tmp/pq
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pq -o p -- python $R/bench.py --no-cpu-baseline --no-extra --traffic off --steps 4 --warmup 2 --launch eager "$@" > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
a=collections.defaultdict(list)
for f in glob.glob("/tmp/pq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KN" in r["Kernel_Name"]: a[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$@", {k: round(sum(v)/len(v)) for k,v in sorted(a.items())})
PY
