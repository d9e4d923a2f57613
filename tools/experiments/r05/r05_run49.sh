#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run49; rm -rf $O; mkdir -p $O
cd $R
( timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "int16 or s16 or clamp" 2>&1 | tail -n 5 ) > $O/s16_tests.log 2>&1; cat $O/s16_tests.log
for k in 1 2 3; do python3 bench.py --workload cfg2 --s16 --no-cpu-baseline --no-host-paths --no-n1-reference 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('cfg2 --s16 %-16s %7.1f us  frac %.3f  parity %s' % (j['roofline']['kernel'], j['ms_per_step']*1e3, j['roofline']['frac'], j['parity_full_stream']))
"; done > $O/s16_pair.log 2>&1; cat $O/s16_pair.log
python3 bench.py --workload cfg2 --no-cpu-baseline --no-host-paths --no-n1-reference 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('cfg2 int32 %7.1f us' % (j['ms_per_step']*1e3))
"
timeout 300 python3 tests/soak_gpu.py --seconds 120 --seed 1401 2>&1 | tail -n 3 | cut -c1-600
