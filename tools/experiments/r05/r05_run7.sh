#!/bin/bash
# k_seg counters (cfg 3) beside k_up2's, + the whole GPU suite on the build so far
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run7; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for leg in seg up2; do
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
             "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS"; do
    i=$((i+1))
    if [ $leg = up2 ]; then export CLOWNRESAMPLER_AMD_NO_SEG=1; else unset CLOWNRESAMPLER_AMD_NO_SEG; fi
    timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_${leg}_$i -- python3 $R/bench.py --workload cfg3 --steps 10 --warmup 2 --no-cpu-baseline --no-check --no-host-paths --no-n1-reference > $O/pmc_${leg}_$i.log 2>&1
    echo "$leg pass $i rc=$?"
  done
  unset CLOWNRESAMPLER_AMD_NO_SEG
  python3 - $O $leg <<'PY'
import csv, glob, sys, collections
out, w = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/pmc_%s_*/**/*counter_collection.csv" % w, recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("k_seg", "k_up")):
            a = agg[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
with open(out + "/counters_%s.txt" % w, "w") as f:
    for k in sorted(agg):
        line = "%-28s per-dispatch mean %.6g over %d dispatches" % (k, agg[k][0] / agg[k][1], agg[k][1])
        print(line); f.write(line + "\n")
PY
  rm -rf $O/pmc_${leg}_[0-9]
done
cd $R
( timeout 1500 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > $O/gpu_tests.log 2>&1
cat $O/gpu_tests.log
