#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run46; rm -rf $O; mkdir -p $O
cd $R && ( timeout 1500 python3 -m pytest tests -m gpu -q --durations=8 2>&1 | tail -n 20 ) > $O/suite.log 2>&1; cat $O/suite.log
