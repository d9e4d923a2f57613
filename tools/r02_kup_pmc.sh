#!/bin/bash
# cfg 3: sustained bench + PMC passes for the given variants (CLOWNRESAMPLER_AMD_VARIANT)
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r02e}; shift
mkdir -p $O
for v in "$@"; do
  export CLOWNRESAMPLER_AMD_VARIANT=$v
  python3 bench.py --workload cfg3 --no-cpu-baseline > $O/bench_cfg3_v$v.json 2> $O/bench_cfg3_v$v.err
  python3 - <<PY
import json
l=json.loads([x for x in open("$O/bench_cfg3_v$v.json") if x.startswith("{")][0])
print("variant $v", l["roofline"]["kernel"], "%.1f us" % (l["ms_per_step"]*1e3), "frac %.3f" % l["roofline"]["frac"], "median %.1f min %.1f" % (l["launch_us"]["median"], l["launch_us"]["min"]))
PY
  bash tools/pmc_passes.sh $O/pmc_v$v --workload cfg3 > $O/pmc_v$v.log 2>&1
  cp $O/pmc_v$v/pmc_summary.txt $O/pmc_summary_cfg3_v$v.txt; rm -rf $O/pmc_v$v
  grep -E "SQ_INSTS_VALU|SQ_ACTIVE_INST_VALU|SQ_LDS_BANK|SQ_LDS_IDX|SQ_WAIT_INST_LDS|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|SQ_WAIT_INST_ANY|SQ_INSTS_SALU|SQ_INSTS_LDS|GRBM_GUI" $O/pmc_summary_cfg3_v$v.txt
done
