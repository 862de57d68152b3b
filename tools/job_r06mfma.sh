#!/bin/bash
# MFMA busy counters of the 4-layer bf16 training step: one PMC pass with --kernel-trace only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06mfma
mkdir -p $OUT
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc -o p -- python $R/tools/train_bench.py --bf16 true --n_layers 4 --steps 8 --hip_graph false > $OUT/train.json 2> $OUT/pmc.err
python3 - <<PY
import csv, glob, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[r["Kernel_Name"]] += 1
rows = []
for k, v in agg.items():
    if v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
        rows.append({"kernel": k[:120], "launches": n[k], "mfma_busy_cycles": v["SQ_VALU_MFMA_BUSY_CYCLES"],
                     "sq_busy_cycles": v.get("SQ_BUSY_CYCLES", 0), "gui_active_cycles": v.get("GRBM_GUI_ACTIVE", 0),
                     "mfma_mops_bf16": v.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0),
                     "mfma_busy_share_of_launch": v["SQ_VALU_MFMA_BUSY_CYCLES"] / max(v.get("GRBM_GUI_ACTIVE", 0) * 128, 1)})
rows.sort(key=lambda r: -r["mfma_busy_cycles"])
tot_busy = sum(r["mfma_busy_cycles"] for r in rows)
tot_act = sum(v.get("GRBM_GUI_ACTIVE", 0) for v in agg.values())
out = {"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES GRBM_GUI_ACTIVE (one PMC pass with --kernel-trace only) over 8 + 5 steps of tools/train_bench.py --bf16 true --n_layers 4 (batch 4, 256x768, eager), summed per kernel. busy share = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128) as in profiles/r05/train_step_mfma_counters.json",
       "kernels": rows[:24], "all_kernels_mfma_busy_share": tot_busy / max(tot_act * 128, 1),
       "kernels_with_mfma": len(rows)}
json.dump(out, open("$OUT/train_step_mfma_counters.json", "w"), indent=1)
print(out["all_kernels_mfma_busy_share"])
for r in rows[:14]: print("%6.3f  %5d  %s" % (r["mfma_busy_share_of_launch"], r["launches"], r["kernel"][:100]))
PY
rm -rf $OUT/pmc
