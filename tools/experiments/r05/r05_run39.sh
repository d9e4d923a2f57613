#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run39; rm -rf $O; mkdir -p $O
cd $R/tools/microbench && timeout 300 ./scatterwrite > $O/scatterwrite.log 2>&1; cat $O/scatterwrite.log
