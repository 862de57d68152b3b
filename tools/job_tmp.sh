cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
rm -rf /tmp/pq
timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pq -o p -- python $R/bench.py --no-cpu-baseline --traffic off --steps 4 --warmup 2 --launch eager > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
a=collections.defaultdict(list)
for f in glob.glob("/tmp/pq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "splat_bwd_kernel" in r["Kernel_Name"] or "splat_stream_kernel<0, true, 1" in r["Kernel_Name"]: a[r["Kernel_Name"][:50]+r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: (round(sum(v)/len(v)), len(v)) for k,v in sorted(a.items())})
PY
done
