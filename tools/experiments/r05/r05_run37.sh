#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run37; rm -rf $O; mkdir -p $O
cd $R
timeout 700 python3 tests/soak_gpu.py --seconds 240 --seed 1101 --trace $O/trace_1101.txt > $O/soak_1101.log 2>&1; echo "rc $?"; tail -n 8 $O/soak_1101.log; tail -n 1 $O/trace_1101.txt
timeout 700 python3 tests/soak_gpu.py --seconds 240 --seed 1102 --abi c99 --trace $O/trace_1102.txt > $O/soak_1102_c99.log 2>&1; echo "rc $?"; tail -n 8 $O/soak_1102_c99.log; tail -n 1 $O/trace_1102.txt
