#!/bin/bash
# Conv path evidence (run on the GPU box): kernel-trace stats of the training
# step in fp32 and bf16, then one PMC pass with the matrix-core counters
# (never combined with other trace domains).   usage: tools/prof_train.sh <outdir>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-train_prof}
mkdir -p $OUT
for bf in false true; do
  python $R/tools/train_bench.py --bf16 $bf --steps 20 > $OUT/train_$bf.json 2> $OUT/train_$bf.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$bf -o kt -- python $R/tools/train_bench.py --bf16 $bf --steps 5 > /dev/null 2> $OUT/kt_$bf.err
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_$bf -o p -- python $R/tools/train_bench.py --bf16 $bf --steps 3 > /dev/null 2> $OUT/pmc_$bf.err
done
python3 - <<PY
import csv, glob, json, collections
out = {}
for bf in ("false", "true"):
    rec = {}
    try:
        rec["bench"] = json.load(open("$OUT/train_%s.json" % bf))
    except Exception as e:
        rec["bench"] = str(e)
    rows = []
    for f in glob.glob("$OUT/kt_%s/**/*kernel_stats.csv" % bf, recursive=True):
        rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    rec["kernel_time_total_ms"] = tot / 1e6
    rec["top_kernels"] = [{"name": r["Name"][:110], "calls": int(r["Calls"]),
                           "total_ms": float(r["TotalDurationNs"]) / 1e6,
                           "pct": 100 * float(r["TotalDurationNs"]) / tot} for r in rows[:25]]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % bf, recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:110]][r["Counter_Name"]] += float(r["Counter_Value"])
    pm = []
    for k, v in agg.items():
        if v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
            pm.append({"name": k, **{c: x for c, x in v.items()}})
    pm.sort(key=lambda r: -r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0))
    rec["mfma_kernels"] = pm[:20]
    rec["sum_counters"] = {c: sum(v.get(c, 0) for v in agg.values()) for c in
                           ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16",
                            "SQ_INSTS_VALU_MFMA_MOPS_F32", "GRBM_GUI_ACTIVE")}
    out["bf16_" + bf] = rec
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
for k, v in out.items():
    print(k, v["bench"])
    print(" kernel time total ms (5+5 warm steps)", round(v["kernel_time_total_ms"], 1))
    for r in v["top_kernels"][:12]:
        print("  %5.1f%% %8.2f ms %5d  %s" % (r["pct"], r["total_ms"], r["calls"], r["name"][:90]))
    print(" counters", v["sum_counters"])
PY
rm -rf $OUT/kt_false $OUT/kt_true $OUT/pmc_false $OUT/pmc_true
