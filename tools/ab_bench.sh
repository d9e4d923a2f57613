#!/bin/bash
# A/B of library builds (tools/ab/lib<NAME>.so) through bench.py (sustained clocks, 200 launches), alternating processes.
#   bash tools/ab_bench.sh "<lib names>" <workload> [<workload> ...]
LIBS=$1; shift
for w in "$@"; do
  for rep in 1 2; do
    for L in $LIBS; do
      CLOWNRESAMPLER_AMD_LIBRARY=$PWD/tools/ab/lib$L.so python bench.py --workload $w --no-cpu-baseline --no-host-paths --no-n1-reference 2>/dev/null | python -c "
import sys,json
l=json.loads(sys.stdin.readline())
print('$w lib$L', l['roofline']['kernel'][:28], 'us %.1f' % (l['ms_per_step']*1e3), 'median %.1f' % l['launch_us']['median'], 'frac %.3f' % l['roofline']['frac'], 'parity', l.get('parity_full_stream'))"
    done
  done
done
