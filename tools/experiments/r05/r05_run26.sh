#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run26; rm -rf $O; mkdir -p $O
cd $R/tools/microbench && timeout 300 ./ldsvalu > $O/ldsvalu.log 2>&1; cat $O/ldsvalu.log
