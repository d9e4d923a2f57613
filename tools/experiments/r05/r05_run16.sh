#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run16; rm -rf $O; mkdir -p $O
cd $R
( timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "segment_kernel or (one_launch and (cfg3 or up6))" 2>&1 | tail -6 ) > $O/seg_tests.log 2>&1
cat $O/seg_tests.log
bash tools/ab_env.sh "-;CLOWNRESAMPLER_AMD_NO_SEG=1;-;CLOWNRESAMPLER_AMD_SEG_TILE=64;CLOWNRESAMPLER_AMD_SEG_TILE=256" cfg3 > $O/seg_ab.log 2>&1
cat $O/seg_ab.log
