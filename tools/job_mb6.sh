cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
MB6_TOUCH=1 ./tools/microbench6.bin > gpurun_out/r03a/microbench6_touch.txt 2>&1
cat gpurun_out/r03a/microbench6_touch.txt
