#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run56; rm -rf $O; mkdir -p $O
cd $R && ( CLOWNRESAMPLER_AMD_DEBUG=1 timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "one_launch and (monob or hq48mb)" 2>&1 | grep -i "dual\|passed\|failed\|error" | tail -n 12 ) > $O/monob.log 2>&1; cat $O/monob.log | cut -c1-260
