set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
python bench.py --workload cfg3 --no-cpu-baseline > gpurun_out/r02a/bench_cfg3.json 2> gpurun_out/r02a/err.log
cat gpurun_out/r02a/bench_cfg3.json
LSI_STREAM_VERBOSE=1 python tools/phase_probe.py cfg3 > gpurun_out/r02a/phase_cfg3.txt 2>&1
cat gpurun_out/r02a/phase_cfg3.txt
bash tools/pmc_quick.sh --workload cfg3 > gpurun_out/r02a/pmc_cfg3.txt 2>&1
cat gpurun_out/r02a/pmc_cfg3.txt
