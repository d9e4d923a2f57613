#!/bin/bash
# round 5, first GPU call: the new parity tests (C99-integer library, HighLevel_Adjust mid-stream), this box's numbers for the shapes
# the round works on, and the LDS-side counters of k_up2 (cfg 3) / k_wave2 (hq44, dn8)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run1; rm -rf $O; mkdir -p $O
cd $R
( timeout 900 python3 -m pytest tests/test_gpu_c99.py tests/test_gpu_parity.py -m gpu -q -k "c99 or adjust" 2>&1 | tail -15 ) > $O/new_tests.log 2>&1
cat $O/new_tests.log
bash tools/ab_env.sh "-" cfg2 cfg3 hq44 hq48 dn8 > $O/baseline.log 2>&1
cat $O/baseline.log
cd /tmp; export TMPDIR=/tmp
for w in cfg3 hq44 dn8; do
  i=0
  for grp in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_WAIT_INST_LDS" \
             "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ATOMIC_RETURN SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" \
             "SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH" \
             "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INSTS_FLAT"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_${w}_$i -- python3 $R/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-check --no-host-paths --no-n1-reference > $O/pmc_${w}_$i.log 2>&1
    echo "$w pass $i rc=$?"
  done
  python3 - $O $w <<'PY'
import csv, glob, sys, collections
out, w = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/pmc_%s_*/**/*counter_collection.csv" % w, recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("k_poly", "k_wave", "k_up", "k_generic", "k_int")):
            a = agg[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
with open(out + "/lds_counters_%s.txt" % w, "w") as f:
    for k in sorted(agg):
        line = "%-28s per-dispatch mean %.6g over %d dispatches" % (k, agg[k][0] / agg[k][1], agg[k][1])
        print(line); f.write(line + "\n")
PY
  tail -3 $O/pmc_${w}_1.log
  rm -rf $O/pmc_${w}_[0-9]
done
