#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run42; rm -rf $O; mkdir -p $O
cd $R && ( timeout 1200 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "one_launch and (cfg3l or cfg2l or hq48l)" --durations=5 2>&1 | tail -n 16 ) > $O/long.log 2>&1; cat $O/long.log
