#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run8; rm -rf $O; mkdir -p $O
cd $R
for rep in 1 2; do for f in 0 5; do CLOWNRESAMPLER_AMD_SEG_FORM=$f python3 bench.py --workload cfg3 --no-check --no-cpu-baseline --no-host-paths --no-n1-reference 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('cfg3 form $f: %7.1f us (median %7.1f)' % (j['ms_per_step']*1e3, j['launch_us']['median']))
    elif 'rror' in l: print(l.strip()[:300])
"; done; done > $O/seg_salu.log 2>&1
cat $O/seg_salu.log
