#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run44; rm -rf $O; mkdir -p $O
cd $R && ( CLOWNRESAMPLER_AMD_DEBUG=1 timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "one_launch and (monol or hq48ml)" --durations=5 2>&1 | grep -v "^clownresampler_amd: plan variant.*threads" | tail -n 30 ) > $O/long.log 2>&1; cat $O/long.log | cut -c1-250
