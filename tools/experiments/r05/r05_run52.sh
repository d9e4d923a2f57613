#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run52; rm -rf $O; mkdir -p $O
cd $R && ( timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "one_launch" 2>&1 | tail -n 4 ) > $O/one_launch.log 2>&1; cat $O/one_launch.log
