#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run51; rm -rf $O; mkdir -p $O
cd $R
for seed in 1501 1502; do
timeout 900 python3 tests/soak_gpu.py --seconds 480 --seed $seed --trace $O/trace_$seed.txt > $O/soak_$seed.log 2>&1; echo "rc $?"; tail -n 4 $O/soak_$seed.log | cut -c1-700
done
