#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run18; rm -rf $O; mkdir -p $O
cd $R
( timeout 1500 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -25 ) > $O/gpu_tests.log 2>&1
cat $O/gpu_tests.log
bash tools/ab_env.sh "-;CLOWNRESAMPLER_AMD_NO_SEG=1" cfg3 > $O/seg_ab.log 2>&1
cat $O/seg_ab.log
