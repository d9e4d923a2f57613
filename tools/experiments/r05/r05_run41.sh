#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run41; rm -rf $O; mkdir -p $O
cd $R && free -g | head -2; ( timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "one_launch and (cfg3h or cfg5)" 2>&1 | tail -n 12 ) > $O/cfg3h.log 2>&1; cat $O/cfg3h.log
