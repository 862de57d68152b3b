#!/bin/bash
# usage: tools/pmc_quick.sh <bench args...>
# Per-launch averages of SQ counters for the splat kernel, 2-3 counters per pass.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY"; do
  rm -rf /tmp/pq
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pq -o p -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 --launch eager "$@" > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
a=collections.defaultdict(list)
for f in glob.glob("/tmp/pq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "splat" in r["Kernel_Name"]: a[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v)/len(v)) for k,v in a.items()})
PY
done
