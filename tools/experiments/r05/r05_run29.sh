#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run29; rm -rf $O; mkdir -p $O
cd $R
( for leg in "-" "CLOWNRESAMPLER_AMD_VARIANT=13" "CLOWNRESAMPLER_AMD_VARIANT=20" "CLOWNRESAMPLER_AMD_VARIANT=29" "CLOWNRESAMPLER_AMD_VARIANT=8" "CLOWNRESAMPLER_AMD_VARIANT=12" "CLOWNRESAMPLER_AMD_TILE_GROUPS=1" "CLOWNRESAMPLER_AMD_TILE_GROUPS=2" "CLOWNRESAMPLER_AMD_TILE_GROUPS=3" "CLOWNRESAMPLER_AMD_DYNAMIC_TILES=0" "-"; do
    [ "$leg" = "-" ] && leg=""
    env $leg python3 bench.py --workload cfg2 --s16 --no-cpu-baseline --no-host-paths --no-n1-reference 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('cfg2 --s16 %-40s %-16s %7.1f us  frac %.3f  parity %s  plan %s' % ('$leg' or '(default)', j['roofline']['kernel'], j['ms_per_step']*1e3, j['roofline']['frac'], j['parity_full_stream'], {k: j['config']['plan'][k] for k in ('kernel','threads','tile_frames','variant','max_blocks')}))
"
  done ) > $O/s16_variants.log 2>&1
cat $O/s16_variants.log
