#!/bin/bash
# Round profile collection (run on the GPU box through gpurun):
#   tools/collect_profiles.sh <round-tag>
# 1. the default bench line (cfg3; with cpu_baseline, the extras and
#    roofline.traffic from bench.py's own two --pmc passes), 2. rocprofv3
#    --kernel-trace --stats of the same command, 3. the other workloads and the
#    per-rank shards of an N-GPU run, 4. SQ counters (PMC passes only, never
#    combined with other trace domains) of the stream and sweep kernels.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for wl in cfg2 cfg5; do
  timeout 300 python $R/bench.py --workload $wl --no-cpu-baseline --no-extra --traffic off > $OUT/bench_$wl.json 2>> $OUT/bench_default.err
done
# (cfg4: with the FETCH / WRITE traffic passes of the any-pose path)
timeout 600 python $R/bench.py --workload cfg4 --no-cpu-baseline --no-extra > $OUT/bench_cfg4.json 2>> $OUT/bench_default.err
for n in 2 4 8; do
  timeout 300 python $R/bench.py --shard-of $n --no-cpu-baseline --no-extra --traffic off > $OUT/bench_cfg3_shard_of_$n.json 2>> $OUT/bench_default.err
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-extra --traffic off > $OUT/kt_bench.json 2> $OUT/kt.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt4 -o kt -- python $R/bench.py --workload cfg4 --no-cpu-baseline --no-extra --traffic off > $OUT/kt4_bench.json 2>> $OUT/kt.err
python3 - <<PY
import csv, glob, json
out = {}
for tag, d in (('default_cfg3', 'kt'), ('cfg4', 'kt4')):
    for f in glob.glob("$OUT/%s/**/*kernel_stats.csv" % d, recursive=True):
        rows = list(csv.DictReader(open(f)))
        out['kernel_stats_' + tag] = [r for r in rows if 'splat' in r['Name'] or 'disp_range' in r['Name']]
json.dump(out, open("$OUT/rocprof_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:2500])
PY
bash $R/tools/pmc_quick.sh splat_stream2_kernel --workload cfg3 > $OUT/pmc_sq_stream_cfg3.txt 2>&1
bash $R/tools/pmc_quick.sh splat_stream2_kernel --workload cfg3 --shard-of 8 > $OUT/pmc_sq_stream_cfg3_shard_of_8.txt 2>&1
bash $R/tools/pmc_quick.sh splat_sweep_kernel --workload cfg4 > $OUT/pmc_sq_sweep_cfg4.txt 2>&1
cat $OUT/pmc_sq_stream_cfg3.txt $OUT/pmc_sq_stream_cfg3_shard_of_8.txt $OUT/pmc_sq_sweep_cfg4.txt
# training-relevant entry points: timings, then counters of their kernels
timeout 300 python $R/tools/time_bwd.py > $OUT/bwd_both_disp_cfg3.json 2>> $OUT/bench_default.err
LSI_BWD_STREAM=0 timeout 300 python $R/tools/time_bwd.py > $OUT/bwd_both_disp_cfg3_gather_kernel.json 2>> $OUT/bench_default.err
timeout 300 python $R/tools/time_bwd.py --workload cfg2 > $OUT/bwd_both_disp_cfg2.json 2>> $OUT/bench_default.err
bash $R/tools/pmc_cmd.sh splat_bwd_stream_kernel python $R/tools/time_bwd.py > $OUT/pmc_bwd_stream_cfg3.txt 2>&1
bash $R/tools/pmc_cmd.sh "768, true" python $R/tools/time_bwd.py > $OUT/pmc_fwd_both_cfg3.txt 2>&1
cat $OUT/bwd_both_disp_cfg3.json $OUT/pmc_bwd_stream_cfg3.txt $OUT/pmc_fwd_both_cfg3.txt
for d in rough stress; do timeout 200 python $R/bench.py --disp $d --no-cpu-baseline --no-extra --traffic off > $OUT/bench_cfg3_$d.json 2>> $OUT/bench_default.err; done
# wall-clock timelines + items by route (stamps build: tools/build_variant.sh stamps lsi_splat_stream2.hip -DS2X_STAMPS)
LSI_HIP_LIB=stamps python $R/tools/phase_probe2.py cfg3 2>&1 | grep -v amdgpu > $OUT/timeline_cfg3.txt
LSI_HIP_LIB=stamps python $R/tools/phase_probe2.py cfg3 0 0 8 2>&1 | grep -v amdgpu > $OUT/timeline_cfg3_shard_of_8.txt
bash $R/tools/pmc_calib.sh > $OUT/pmc_calibration.txt 2>&1
rm -rf $OUT/kt/*/ $OUT/kt4/*/ 2>/dev/null
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/bench_default.json
lscpu | grep -E "Model name|^CPU\(s\)|Socket" > $OUT/host.txt; cat $OUT/host.txt
