#!/bin/bash
# round 6: conv / train / nets GPU tests after the split over the input channels,
# per-layer timing with and without it
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06c
mkdir -p $OUT
timeout 800 python -m pytest tests/test_conv_gpu.py tests/test_train_gpu.py tests/test_nets_golden.py -q -m gpu -n 4 2>&1 | tail -120 > $OUT/conv_tests.log
for t in 0 512 768 1024; do
  LSI_IGEMM_SPLITK=$t timeout 600 python tools/conv_bench.py --out $OUT/conv_bench_splitk$t.json > $OUT/conv_bench_splitk$t.txt 2>&1
done
tail -4 $OUT/conv_tests.log; for t in 0 512 768 1024; do tail -1 $OUT/conv_bench_splitk$t.txt; done
