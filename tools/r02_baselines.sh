#!/bin/bash
# round-2 starting points for the VALU-bound shapes (one box, sustained clocks)
O=gpurun_out/${1:-r02c}
mkdir -p $O
python tools/sweep_variants.py --workload hq48 --variants 13,20,21,28,29 > $O/sweep_hq48.log 2>&1; tail -12 $O/sweep_hq48.log
python tools/sweep_variants.py --workload cfg3 --variants 20,26,27 > $O/sweep_cfg3.log 2>&1; tail -8 $O/sweep_cfg3.log
for w in cfg3 hq48 hq44 dn8 dn2 up55 cfg4; do
  python bench.py --workload $w --no-cpu-baseline --no-check > $O/bench_$w.json 2> $O/bench_$w.err
  python - <<PY
import json
l=json.loads([x for x in open("$O/bench_$w.json") if x.startswith("{")][0])
print("$w", l["roofline"]["kernel"], "%.1f us" % (l["ms_per_step"]*1e3), "frac %.3f" % l["roofline"]["frac"], l["launch_us"]["median"], l["launch_us"]["min"])
PY
done
