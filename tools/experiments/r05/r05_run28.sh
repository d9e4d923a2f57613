#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run28; rm -rf $O; mkdir -p $O
cd $R
export CLOWNRESAMPLER_AMD_ROTATE_MIN_GAIN=0
for w in cfg2; do
  for f in 0 1 2 3 4 5 6; do
    CLOWNRESAMPLER_AMD_W2_FORM=$f timeout 300 python3 bench.py --workload $w --no-check --no-cpu-baseline --no-host-paths --no-n1-reference 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('%-6s rows rotated, k_poly form $f (1: window reads conflict-free, 2: row reads, 3: both, 4: no DMA no stores, 5: 3 + 4, 6: no stores): %7.1f us  frac %.3f  %s' % ('$w', j['ms_per_step']*1e3, j['roofline']['frac'], j['roofline']['kernel']))
"
  done
done > $O/kpoly_forms.log 2>&1
cat $O/kpoly_forms.log
unset CLOWNRESAMPLER_AMD_ROTATE_MIN_GAIN
( for leg in "-" "CLOWNRESAMPLER_AMD_ROTATE_MIN_GAIN=0" "-" "CLOWNRESAMPLER_AMD_ROTATE_MIN_GAIN=0"; do
    [ "$leg" = "-" ] && leg=""
    for extra in "--s16" ""; do
    env $leg python3 bench.py --workload cfg2 $extra --no-cpu-baseline --no-host-paths --no-n1-reference 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('cfg2 %-6s %-40s %-16s %7.1f us  frac %.3f  parity %s' % ('$extra', '$leg' or '(default: rows plain)', j['roofline']['kernel'], j['ms_per_step']*1e3, j['roofline']['frac'], j['parity_full_stream']))
"
    done
  done ) > $O/s16_rotation_ab.log 2>&1
cat $O/s16_rotation_ab.log
