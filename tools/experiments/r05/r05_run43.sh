#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run43; rm -rf $O; mkdir -p $O
cd $R && ( timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "one_launch and (cfg4l or dn6xl or dn8l)" --durations=5 2>&1 | tail -n 30 ) > $O/long.log 2>&1; cat $O/long.log
