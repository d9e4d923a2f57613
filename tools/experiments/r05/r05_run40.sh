#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run40; rm -rf $O; mkdir -p $O
cd $R && timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -n 5 $O/smoke.log
timeout 300 python3 bench.py --steps 20 --warmup 3 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['metric'], j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('traffic'), j['cpu_baseline'])
"
