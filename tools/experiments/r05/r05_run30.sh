#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run30; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python3 tests/soak_gpu.py --seconds 420 --seed 101 > $O/soak_101.log 2>&1; tail -30 $O/soak_101.log
timeout 500 python3 tests/soak_gpu.py --seconds 240 --seed 202 > $O/soak_202.log 2>&1; tail -30 $O/soak_202.log
