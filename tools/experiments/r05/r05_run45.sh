#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run45; rm -rf $O; mkdir -p $O
cd $R
for seed in 1301 1302 1303; do
timeout 800 python3 tests/soak_gpu.py --seconds 420 --seed $seed --trace $O/trace_$seed.txt > $O/soak_$seed.log 2>&1; echo "rc $?"; tail -n 6 $O/soak_$seed.log | cut -c1-700; tail -n 1 $O/trace_$seed.txt
done
timeout 800 python3 tests/soak_gpu.py --seconds 300 --seed 1304 --abi c99 --trace $O/trace_1304.txt > $O/soak_1304_c99.log 2>&1; echo "rc $?"; tail -n 6 $O/soak_1304_c99.log | cut -c1-700
