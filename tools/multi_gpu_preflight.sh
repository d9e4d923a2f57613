#!/bin/bash
# multi_gpu_preflight.sh - ONE command for the day a node with several MI355X is at hand.
#
#   bash tools/multi_gpu_preflight.sh [outdir]            on a node with >= 2 GPUs: the real thing (RCCL over xGMI, peer copies)
#   CRA_PREFLIGHT_DRY=1 bash tools/multi_gpu_preflight.sh the steps a ONE-GPU box allows (ranks share the GPU over gloo; validation
#                                                         only: the timings are those of ranks contending for one GPU)
#   CRA_PREFLIGHT_DRY=1 CRA_PREFLIGHT_QUICK=1 ...         the dry run in two minutes, for the GPU test suite to run on every box
#                                                         (tests/test_gpu_devices.py::test_preflight_dry_run): no pytest step (the suite
#                                                         is the caller), a minute of audio over 8 shards, bench at 1 and 2 ranks, few steps
#
# What has never run on distinct GPUs and runs here, in this order, every step in a FRESH process under its
# own timeout (never an exec from a process that has touched the GPU; a hung step is killed and reported, the rest still runs):
#   1. pytest: two rank processes on two distinct GPUs over RCCL; the per-device contexts; the one-call sharded entry point
#   2. tools/bin/cr_multi: one process, every GPU, ONE C call (cfg 5's hour), concatenate by peer copies, then by RCCL
#      (CLOWNRESAMPLER_AMD_GATHER_RCCL stays EXPERIMENTAL until this step has passed on real hardware)
#   3. bench.py --gpus 1 / 2 / 4 / 8: one JSON line each into <outdir>, and a table: kernel-only Msamples/s, gather-to-root,
#      all-gather, efficiency_vs_n1, link GB/s
# Writes <outdir>/preflight.log (default profiles/multi_gpu_preflight/); returns 0 only if every step that ran passed.
set -u
cd "$(dirname "$0")/.."
ROOT=$(pwd)
OUT=${1:-$ROOT/profiles/multi_gpu_preflight}
mkdir -p "$OUT"
LOG=$OUT/preflight.log
: > "$LOG"
export HSA_ENABLE_IPC_MODE_LEGACY=0
DRY=${CRA_PREFLIGHT_DRY:-0}
QUICK=${CRA_PREFLIGHT_QUICK:-0}
[ "$QUICK" = 1 ] && DRY=1
BENCH_ARGS=$([ "$QUICK" = 1 ] && echo "--steps 6 --warmup 2 --prewarm-ms 10" || echo "")
say() { echo "$@" | tee -a "$LOG"; }
FAILED=0
step() {   # step <seconds> <name> <command...>
	local limit=$1 name=$2; shift 2
	say "== $name (limit ${limit}s): $*"
	local t0=$(date +%s)
	timeout --kill-after=20 "$limit" "$@" >> "$LOG" 2>&1
	local rc=$?
	say "-- $name: rc=$rc in $(( $(date +%s) - t0 ))s$( [ $rc -eq 124 ] && echo ' (TIMED OUT)' )"
	[ $rc -ne 0 ] && FAILED=1
	return $rc
}

# how many GPUs: counted without initialising any of them in this shell
NGPU=$(python3 -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
say "multi_gpu_preflight: $NGPU GPU(s) visible, dry=$DRY, $(date -u +%FT%TZ), library source id $(python3 -c 'import sys; sys.path.insert(0, "."); import clownresampler_amd as cr; print(cr.load(3).BuildId())' 2>/dev/null)"
if [ "$NGPU" -lt 2 ] && [ "$DRY" != 1 ]; then
	say "fewer than two GPUs: nothing here can run for real.  CRA_PREFLIGHT_DRY=1 runs the shared-GPU validation of steps 1-3."
	exit 2
fi

# ---- 1. the tests --------------------------------------------------------------------------------------------------------
if [ "$QUICK" = 1 ]; then
	say "== tests: skipped in a quick dry run (the test suite is the caller)"
elif [ "$DRY" = 1 ]; then
	step 900 "tests (shared GPU: rank processes + device contexts)" python3 -m pytest tests/test_gpu_ranks.py tests/test_gpu_devices.py -m gpu -q -x
else
	step 600 "tests: two distinct GPUs over RCCL" python3 -m pytest "tests/test_gpu_ranks.py::test_two_distinct_gpus_over_rccl" -m gpu -q -x
	step 900 "tests: rank processes, device contexts, sharded call" python3 -m pytest tests/test_gpu_ranks.py tests/test_gpu_devices.py -m gpu -q -x
fi

# ---- 2. one process, every device, one C call ------------------------------------------------------------------------------
HOUR=158760000    # BASELINE configs[4]: one hour of stereo 44.1 kHz
SHARDS=$([ "$NGPU" -ge 8 ] && echo 8 || echo "$NGPU"); [ "$DRY" = 1 ] && SHARDS=8
FRAMES=$([ "$DRY" = 1 ] && echo 26460000 || echo $HOUR)
[ "$QUICK" = 1 ] && FRAMES=2646000
step 900 "cr_multi $SHARDS shards, peer copies" tools/bin/cr_multi "$SHARDS" "$FRAMES" peer
if [ "$DRY" = 1 ]; then
	say "== cr_multi rccl: skipped in a dry run (with one device no RCCL operation is issued: nothing would be learnt)"
else
	step 900 "cr_multi $SHARDS shards, RCCL gather (EXPERIMENTAL until this passes)" env CLOWNRESAMPLER_AMD_EXPERIMENTAL_RCCL=1 tools/bin/cr_multi "$SHARDS" "$FRAMES" rccl
fi

# ---- 3. the curve ----------------------------------------------------------------------------------------------------------
for n in 1 2 4 8; do
	if [ "$QUICK" = 1 ] && [ "$n" -gt 2 ]; then continue; fi
	if [ "$DRY" != 1 ] && [ "$n" -gt "$NGPU" ]; then say "== bench --gpus $n: skipped ($NGPU GPUs)"; continue; fi
	port=$((29500 + n))
	if [ "$n" = 1 ]; then
		step 900 "bench --gpus 1" bash -c "python3 bench.py --gpus 1 $BENCH_ARGS > '$OUT/bench_n1.json' 2> '$OUT/bench_n1.err'"
	elif [ "$DRY" = 1 ]; then
		step 1200 "bench --gpus $n (ranks share one GPU, gloo: VALIDATION ONLY)" bash -c "CRA_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n --rank-timeout 900 $BENCH_ARGS > '$OUT/bench_n$n.json' 2> '$OUT/bench_n$n.err'"
	else
		step 1200 "bench --gpus $n (RCCL)" bash -c "python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n --rank-timeout 900 > '$OUT/bench_n$n.json' 2> '$OUT/bench_n$n.err'"
	fi
done

python3 - "$OUT" <<'PY' | tee -a "$LOG"
import glob, json, os, sys
out = sys.argv[1]
rows = []
for path in sorted(glob.glob(os.path.join(out, "bench_n*.json")), key=lambda p: int(p.rsplit("_n", 1)[1].split(".")[0])):
    line = next((l for l in open(path) if l.startswith("{")), None)
    if line is None:
        print("%s: no JSON line (see %s)" % (os.path.basename(path), path.replace(".json", ".err")))
        continue
    j = json.loads(line)
    g = j.get("gather") or {}
    eff = j.get("efficiency_vs_n1")
    rows.append((j["n_gpus"], j["value"], j["ms_per_step"] * 1e3, eff.get("value") if isinstance(eff, dict) else eff, (g.get("to_root") or {}).get("ms"),
                 (g.get("all_gather") or {}).get("ms"), g.get("link_GBs"), j.get("backend", "-"), j.get("distinct_devices"), j.get("parity_full_stream")))
print("%5s %14s %10s %10s %12s %13s %9s %9s %7s  %s" % ("gpus", "Msamples/s", "us/step", "eff vs n1", "to root ms", "all-gather ms", "link GB/s", "devices", "parity", "backend"))
for r in rows:
    f = lambda v, spec: ("%" + spec) % v if isinstance(v, (int, float)) else "-"
    print("%5d %14.0f %10.1f %10s %12s %13s %9s %9s %7s  %s" % (r[0], r[1], r[2], f(r[3], ".3f"), f(r[4], ".2f"), f(r[5], ".2f"), f(r[6], ".1f"), f(r[8], "d"), r[9], r[7]))
PY
say "multi_gpu_preflight: $([ $FAILED = 0 ] && echo 'ALL STEPS PASSED' || echo 'SOME STEP FAILED - see above')"
exit $FAILED
