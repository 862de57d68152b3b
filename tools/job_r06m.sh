#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06m
mkdir -p $OUT
( time timeout 1500 python -m pytest tests/ -q -m gpu 2>&1 | tail -30 ) > $OUT/gputest.log 2>&1
( time timeout 900 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
tail -4 $OUT/gputest.log; tail -2 $OUT/smoke.log; cut -c1-400 $OUT/bench_default.json
