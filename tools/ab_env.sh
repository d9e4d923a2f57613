#!/bin/bash
# A/B on ONE box: bench.py workloads under environment settings.  usage: ab_env.sh "<VAR=val ...|->;<...>" w1 w2 ...
IFS=';' read -ra LEGS <<< "$1"; shift
for w in "$@"; do
  for leg in "${LEGS[@]}"; do
    [ "$leg" = "-" ] && leg=""
    env $leg python3 bench.py --workload $w --no-cpu-baseline --no-host-paths --no-n1-reference 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('%-8s %-40s %-16s %7.1f us  frac %.3f  parity %s' % ('$w', '$leg' or '(default)', j['roofline']['kernel'], j['ms_per_step']*1e3, j['roofline']['frac'], j['parity_full_stream']))
    elif 'rror' in l or 'differs' in l or 'bench:' in l: print(l.strip()[:300])
"
  done
done
