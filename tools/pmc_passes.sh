#!/bin/bash
# rocprofv3 PMC passes over bench.py (one counter group per run; never combined with tracing).  Run on the GPU box:
#   bash tools/pmc_passes.sh <outdir> [bench args...]
set -u
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_COUNT" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc$i" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-check --no-host-paths --no-n1-reference "$@" > "$OUT/pmc$i.log" 2>&1
  echo "pass $i ($grp): rc=$?"
done
python3 - "$OUT" "$GRAFT_REPO_ROOT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
sys.path.insert(0, sys.argv[2])
import clownresampler_amd as cr
build_id = cr.load(3).BuildId()
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("k_poly", "k_wave", "k_up", "k_seg", "k_generic", "k_int")):
            a = agg[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
with open(out + "/pmc_summary.txt", "w") as w:
    # the build these counters belong to: bench.py quotes the summary only for the library with this source id
    w.write("library_source_id %s\n" % build_id)
    for k in sorted(agg):
        line = "%-28s per-dispatch mean %.6g over %d dispatches" % (k, agg[k][0] / agg[k][1], agg[k][1])
        print(line); w.write(line + "\n")
PY
