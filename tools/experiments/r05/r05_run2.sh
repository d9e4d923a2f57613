#!/bin/bash
# round 5, GPU call 2: the high-level tests after the pull-boundary fix + k_up2 timing ablations (rows once / conflict-free staging)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run2; rm -rf $O; mkdir -p $O
cd $R
( timeout 900 python3 -m pytest tests/test_gpu_c99.py tests/test_gpu_parity.py -m gpu -q -k "c99 or adjust or highlevel" 2>&1 | tail -15 ) > $O/new_tests.log 2>&1
cat $O/new_tests.log
python3 tools/sweep_variants.py --workload cfg3 --variants 27,1009,1010,1011,1012,1013 > $O/kup2_ablations.log 2>&1
cat $O/kup2_ablations.log
