#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run32; rm -rf $O; mkdir -p $O
cd $R
for seed in 101 202 303; do
timeout 700 python3 tests/soak_gpu.py --seconds 300 --seed $seed --trace $O/trace_$seed.txt > $O/soak_$seed.log 2>&1; echo "rc $?"; tail -12 $O/soak_$seed.log; tail -1 $O/trace_$seed.txt
done
