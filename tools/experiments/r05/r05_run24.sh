#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run24; rm -rf $O; mkdir -p $O
cd $R
( cd tools/microbench && timeout 300 ./ldswin ) > $O/ldswin.log 2>&1; cat $O/ldswin.log
for w in hq44 dn8 hq48; do
  for f in 0 1 2 3; do
    CLOWNRESAMPLER_AMD_W2_FORM=$f timeout 300 python3 bench.py --workload $w --no-check --no-cpu-baseline --no-host-paths --no-n1-reference 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('%-6s k_wave2 form $f (1: window reads conflict-free, 2: row reads, 3: both): %7.1f us  frac %.3f  %s' % ('$w', j['ms_per_step']*1e3, j['roofline']['frac'], j['roofline']['kernel']))
"
  done
done > $O/w2_forms.log 2>&1
cat $O/w2_forms.log
