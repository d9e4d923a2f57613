#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run36; rm -rf $O; mkdir -p $O
cd $R
( timeout 900 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -6 ) > $O/suite.log 2>&1; cat $O/suite.log
for seed in 1001 1002; do
timeout 700 python3 tests/soak_gpu.py --seconds 300 --seed $seed --trace $O/trace_$seed.txt > $O/soak_$seed.log 2>&1; echo "rc $?"; tail -n 8 $O/soak_$seed.log; tail -n 1 $O/trace_$seed.txt
done
