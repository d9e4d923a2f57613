#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run3; rm -rf $O; mkdir -p $O
cd $R
tools/microbench/pkfma_sgpr > $O/pkfma_sgpr.log 2>&1; cat $O/pkfma_sgpr.log
( timeout 900 python3 -m pytest tests/test_gpu_c99.py tests/test_gpu_parity.py -m gpu -q -k "c99 or adjust or highlevel" 2>&1 | tail -15 ) > $O/new_tests.log 2>&1
cat $O/new_tests.log
