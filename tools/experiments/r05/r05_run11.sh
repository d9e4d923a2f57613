#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run11; rm -rf $O; mkdir -p $O
cd $R
( timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "segment_kernel or (one_launch and (cfg3 or hq48))" 2>&1 | tail -12 ) > $O/seg_tests.log 2>&1
cat $O/seg_tests.log
bash tools/ab_env.sh "-;CLOWNRESAMPLER_AMD_NO_SEG=1;CLOWNRESAMPLER_AMD_SEG_TILE=64;CLOWNRESAMPLER_AMD_SEG_TILE=256" cfg3 hq48 > $O/seg_ab.log 2>&1
cat $O/seg_ab.log
for w in cfg3 hq48; do for f in 1 2 3; do CLOWNRESAMPLER_AMD_SEG_FORM=$f python3 bench.py --workload $w --no-check --no-cpu-baseline --no-host-paths --no-n1-reference 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$w ablation form $f: %7.1f us' % (j['ms_per_step']*1e3))
    elif 'rror' in l: print(l.strip()[:300])
"; done; done > $O/seg_ablations.log 2>&1
cat $O/seg_ablations.log
