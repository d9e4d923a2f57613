#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run33; rm -rf $O; mkdir -p $O
cd $R
for seed in 404 505; do
timeout 700 python3 tests/soak_gpu.py --seconds 360 --seed $seed --trace $O/trace_$seed.txt > $O/soak_$seed.log 2>&1; echo "rc $?"; tail -12 $O/soak_$seed.log; tail -1 $O/trace_$seed.txt
done
timeout 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "soak" 2>&1 | tail -5
