cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03am
MB6_STAGGER=1 ./tools/microbench6.bin > gpurun_out/r03am/microbench6_stagger.txt 2>&1
cat gpurun_out/r03am/microbench6_stagger.txt
