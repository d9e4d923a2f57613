#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run25; rm -rf $O; mkdir -p $O
cd $R
( timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_c99.py -m gpu -q -x 2>&1 | tail -15 ) > $O/tests.log 2>&1
cat $O/tests.log
for w in hq48 hq44 dn8 hq48m hq44m dn8m; do
  for f in 0 4; do
    CLOWNRESAMPLER_AMD_W2_FORM=$f timeout 300 python3 bench.py --workload $w --no-cpu-baseline --no-host-paths --no-n1-reference 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('%-6s k_wave2 form $f (0: frames pipelined, 4: frame by frame): %7.1f us  frac %.3f  %s parity %s' % ('$w', j['ms_per_step']*1e3, j['roofline']['frac'], j['roofline']['kernel'], j['parity_full_stream']))
"
  done
done > $O/w2_pipe.log 2>&1
cat $O/w2_pipe.log
