#!/bin/bash
# tools/pmc_calib.sh: FETCH_SIZE / WRITE_SIZE per access pattern on a known byte
# count (tools/pmc_calib.hip), one counter per PMC-only pass; prints
# counter_KiB * 1024 / bytes -- the factor to divide a kernel's counter by.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
[ -x $R/tools/pmc_calib.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/tools/pmc_calib.bin $R/tools/pmc_calib.hip
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pcal
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pcal -o p -- $R/tools/pmc_calib.bin > /tmp/pcal.out 2>&1
  python3 - <<PY
import csv, glob, collections
bytes_ = 768 << 20
a = collections.defaultdict(list)
for f in glob.glob("/tmp/pcal/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("calib_"): a[r["Kernel_Name"]].append(float(r["Counter_Value"]))
print("$c  (counter KiB * 1024 / bytes moved per direction; %d MiB per launch)" % (bytes_ >> 20))
for k in sorted(a):
    v = a[k][-1]
    print("  %-24s %12.0f KiB   factor %.4f" % (k, v, v * 1024.0 / bytes_))
PY
done
