#!/bin/bash
# usage: tools/pmc_quick.sh <kernel-name-substring> <bench args...>
# Per-launch averages of SQ counters for one kernel, a few counters per pass
# (PMC passes only: never combined with other trace domains).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
KN=$1; shift
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  rm -rf /tmp/pq
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pq -o p -- python $R/bench.py --no-cpu-baseline --no-extra --traffic off --steps 4 --warmup 2 --launch eager "$@" > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
a=collections.defaultdict(list)
for f in glob.glob("/tmp/pq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KN" in r["Kernel_Name"]: a[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v)/len(v)) for k,v in a.items()}, 'launches', {k: len(v) for k,v in a.items()}.popitem()[1] if a else 0)
PY
done
