#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run14; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python3 tools/seg_ratio_sweep.py > $O/seg_ratio_sweep.log 2>&1; cat $O/seg_ratio_sweep.log
