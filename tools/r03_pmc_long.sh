#!/bin/bash
# Round 3, step 1 of VERDICT r2 item 1: counter evidence for the long downsampling windows BEFORE any kernel change.
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03pmc}
shift || true
W=${@:-dn8 hq44 dn6x dn6xm}
rm -rf $O; mkdir -p $O
for w in $W; do
  python3 $R/bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
  bash $R/tools/pmc_passes.sh $O/pmc_$w --workload $w > $O/pmc_$w.log 2>&1
  cp $O/pmc_$w/pmc_summary.txt $O/pmc_summary_$w.txt
  rm -rf $O/pmc_$w
  python3 - <<PY
import json
l=json.loads([x for x in open("$O/bench_$w.json") if x.startswith("{")][0])
print("$w", l["roofline"]["kernel"], "%.1f us" % (l["ms_per_step"]*1e3), "frac %.3f" % l["roofline"]["frac"], l["config"]["plan"])
PY
  cat $O/pmc_summary_$w.txt
done
