#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run23; rm -rf $O; mkdir -p $O
cd $R
( timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "segment_kernel or (one_launch and (cfg3 or up6 or hq48))" 2>&1 | tail -12 ) > $O/seg_tests.log 2>&1
cat $O/seg_tests.log
bash tools/ab_env.sh "-;CLOWNRESAMPLER_AMD_NO_SEG=1;-;CLOWNRESAMPLER_AMD_SEG_TILE=128" hq48 cfg3 > $O/seg_ab.log 2>&1
cat $O/seg_ab.log
timeout 900 python3 tools/seg_ratio_sweep.py 44100:48000 32000:44100 32000:48000 22050:44100 22050:48000 16000:44100 16000:48000 11025:44100 8000:36000 > $O/seg_ratio_sweep_low.log 2>&1; cat $O/seg_ratio_sweep_low.log
