#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run10; rm -rf $O; mkdir -p $O
cd $R
bash tools/ab_env.sh "-;CLOWNRESAMPLER_AMD_SEG_TILE=32;CLOWNRESAMPLER_AMD_SEG_FORM=6;CLOWNRESAMPLER_AMD_SEG_TILE=32;CLOWNRESAMPLER_AMD_SEG_FORM=6" cfg3 > $O/seg_ab.log 2>&1
cat $O/seg_ab.log
