#!/bin/bash
# Round profile collection (run on the GPU box through gpurun):
#   tools/collect_profiles.sh <round-tag>
# 1. the default bench line (with cpu_baseline), 2. rocprofv3 --kernel-trace
# --stats of the same command, 3. FETCH_SIZE and WRITE_SIZE in separate PMC
# passes (never combined with other trace domains), 4. other workloads.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for wl in cfg3 cfg4 cfg5; do
  python $R/bench.py --workload $wl --no-cpu-baseline > $OUT/bench_$wl.json 2>> $OUT/bench_default.err
done
python $R/bench.py --launch eager --no-cpu-baseline > $OUT/bench_default_eager.json 2>> $OUT/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $R/bench.py --no-cpu-baseline > $OUT/kt_bench.json 2> $OUT/kt.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --launch eager > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o w -- python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --launch eager > /dev/null 2> $OUT/write.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch3 -o f -- python $R/bench.py --workload cfg3 --no-cpu-baseline --steps 20 --warmup 5 --launch eager > /dev/null 2>> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write3 -o w -- python $R/bench.py --workload cfg3 --no-cpu-baseline --steps 20 --warmup 5 --launch eager > /dev/null 2>> $OUT/write.err
python3 - <<PY
import csv, glob, json, collections, re
out = {}
def kstats(path):
    rows = list(csv.DictReader(open(path)))
    return [r for r in rows if 'splat' in r['Name']]
for f in glob.glob("$OUT/kt/*kernel_stats.csv"):
    out['kernel_stats'] = kstats(f)
def pmc(dirname, counter):
    res = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % dirname):
        for row in csv.DictReader(open(f)):
            if 'splat' in row['Kernel_Name'] and row['Counter_Name'] == counter:
                name = re.search(r'splat_\w+(<[^>]*>)?', row['Kernel_Name']).group(0)
                res[name].append(float(row['Counter_Value']))
    return {k: {'n': len(v), 'mean': sum(v)/len(v), 'min': min(v), 'max': max(v)} for k, v in res.items()}
out['FETCH_SIZE_cfg2'] = pmc('fetch', 'FETCH_SIZE'); out['WRITE_SIZE_cfg2'] = pmc('write', 'WRITE_SIZE')
out['FETCH_SIZE_cfg3'] = pmc('fetch3', 'FETCH_SIZE'); out['WRITE_SIZE_cfg3'] = pmc('write3', 'WRITE_SIZE')
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
cat $OUT/bench_default.json
lscpu | grep -E "Model name|^CPU\(s\)|Socket" > $OUT/host.txt; cat $OUT/host.txt
