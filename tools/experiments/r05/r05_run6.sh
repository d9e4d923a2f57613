#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run6; rm -rf $O; mkdir -p $O
cd $R
tools/microbench/scatterwrite > $O/scatterwrite.log 2>&1; cat $O/scatterwrite.log
bash tools/ab_env.sh "CLOWNRESAMPLER_AMD_SEG_TILE=32;CLOWNRESAMPLER_AMD_SEG_TILE=64;CLOWNRESAMPLER_AMD_SEG_TILE=128" cfg3 > $O/seg_tiles.log 2>&1
cat $O/seg_tiles.log
for t in 64 128; do for f in 1 2 3; do CLOWNRESAMPLER_AMD_SEG_TILE=$t CLOWNRESAMPLER_AMD_SEG_FORM=$f python3 bench.py --workload cfg3 --no-check --no-cpu-baseline --no-host-paths --no-n1-reference 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('cfg3 tile $t ablation form $f: %7.1f us' % (j['ms_per_step']*1e3))
    elif 'rror' in l: print(l.strip()[:300])
"; done; done > $O/seg_ablations.log 2>&1
cat $O/seg_ablations.log
