#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run47; rm -rf $O; mkdir -p $O
cd $R && bash tools/ab_env.sh "-;-" hq44 hq44m > $O/hq44_12waves.log 2>&1; cat $O/hq44_12waves.log
