#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run31; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python3 tests/soak_gpu.py --seconds 420 --seed 101 --trace $O/trace_101.txt > $O/soak_101.log 2>&1; tail -5 $O/soak_101.log; tail -2 $O/trace_101.txt; wc -l $O/trace_101.txt
timeout 500 python3 tests/soak_gpu.py --seconds 240 --seed 202 --trace $O/trace_202.txt > $O/soak_202.log 2>&1; tail -5 $O/soak_202.log; tail -2 $O/trace_202.txt; wc -l $O/trace_202.txt
