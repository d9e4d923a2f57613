#!/bin/bash
# PMC passes + bench for one workload under given variants: bash tools/r02_pmc_variant.sh <outdir-name> <workload> <variant>...
O=$GRAFT_REPO_ROOT/gpurun_out/$1; W=$2; shift; shift
mkdir -p $O
for v in "$@"; do
  export CLOWNRESAMPLER_AMD_VARIANT=$v
  python3 bench.py --workload $W --no-cpu-baseline > $O/bench_${W}_v$v.json 2> $O/bench_${W}_v$v.err
  python3 - <<PY
import json
l=json.loads([x for x in open("$O/bench_${W}_v$v.json") if x.startswith("{")][0])
print("$W variant $v", l["roofline"]["kernel"], "%.1f us" % (l["ms_per_step"]*1e3), "frac %.3f" % l["roofline"]["frac"], "median %.1f min %.1f" % (l["launch_us"]["median"], l["launch_us"]["min"]))
PY
  bash tools/pmc_passes.sh $O/pmc_v$v --workload $W > $O/pmc_${W}_v$v.log 2>&1
  cp $O/pmc_v$v/pmc_summary.txt $O/pmc_summary_${W}_v$v.txt; rm -rf $O/pmc_v$v
  grep -E "SQ_INSTS_VALU|SQ_ACTIVE_INST_VALU|SQ_LDS_BANK|SQ_LDS_IDX|SQ_WAIT_INST_LDS|SQ_WAVE_CYCLES|SQ_WAIT_INST_ANY|SQ_INSTS_LDS|GRBM_GUI|SQ_ACTIVE_INST_LDS|SQ_ACTIVE_INST_ANY" $O/pmc_summary_${W}_v$v.txt
done
