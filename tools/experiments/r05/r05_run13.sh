#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run13; rm -rf $O; mkdir -p $O
cd $R
python3 tools/kseg_phases.py cfg3 > $O/phases_cfg3.log 2>&1; cat $O/phases_cfg3.log
python3 tools/kseg_phases.py hq48 > $O/phases_hq48.log 2>&1; cat $O/phases_hq48.log
