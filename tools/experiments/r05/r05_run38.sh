#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run38; rm -rf $O; mkdir -p $O
cd $R
timeout 700 python3 tests/soak_gpu.py --seconds 300 --seed 1201 --trace $O/trace_1201.txt > $O/soak_1201.log 2>&1; echo "rc $?"; tail -n 8 $O/soak_1201.log | cut -c1-700; tail -n 1 $O/trace_1201.txt
