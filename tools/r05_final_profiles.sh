#!/bin/bash
# Round 5 evidence, one box, ONE library build (the PMC summaries are stamped with its source id; bench.py quotes them only for that build):
# the GPU test suite; bench lines + rocprofv3 kernel traces for the BASELINE configurations, with the mean over the TIMED dispatches taken
# from the trace (tools/trace_timed_mean.py: VERDICT r4 item 5); PMC passes for them and for the LDS-bound long-window shapes (hq48,
# hq44, dn8); every workload of bench.py's table; k_seg's A/B, ablations, phases and ratio sweep; the N > 1 validation runs on the one
# GPU; the channel tables; the host paths.
#   gpurun --timeout 3300 -- 'bash tools/r05_final_profiles.sh'     then     python tools/r05_install_evidence.py
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05final
rm -rf $O; mkdir -p $O
( cd $R; timeout 1500 python3 -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/gpu_tests.log 2>&1
tail -2 $O/gpu_tests.log
for w in cfg2 cfg3 cfg4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -- python3 $R/bench.py --workload $w --no-cpu-baseline --no-host-paths --no-n1-reference > $O/trace_$w.json 2> $O/trace_$w.err
  f=$(find $O/trace_$w -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$w.csv
  t=$(find $O/trace_$w -name "*kernel_trace.csv" | head -1); head -6 "$t" > $O/kernel_trace_head_$w.csv
  python3 $R/tools/trace_timed_mean.py "$t" $O/trace_$w.json >> $O/trace_timed_means.log 2>&1
  rm -rf $O/trace_$w $O/trace_$w.err
done
cat $O/trace_timed_means.log
for w in cfg2 cfg3 cfg4 hq48 hq44 dn8; do
  bash $R/tools/pmc_passes.sh $O/pmc_$w --workload $w > $O/pmc_$w.log 2>&1
  cp $O/pmc_$w/pmc_summary.txt $O/pmc_summary_$w.txt
  rm -rf $O/pmc_$w $O/pmc_$w.log
done
cd $R
# the bench lines, now that the stamped summaries exist (`traffic` and `roofline_valu` filled in for the BASELINE configurations)
for w in cfg2 cfg3 cfg4 hq48 hq44 dn8; do cp $O/pmc_summary_$w.txt profiles/r05_${w}_pmc_summary.txt; done
python3 bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python3 bench.py --steps 20 --warmup 3 > $O/bench_cfg2_steps20.json 2>/dev/null     # as the driver runs it
for w in cfg3 cfg4; do python3 bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; done
for w in cfg5 hq48 hq44 dn8; do python3 bench.py --workload $w --no-cpu-baseline --no-host-paths > $O/bench_$w.json 2>/dev/null; done
python3 bench.py --workload cfg2 --s16 --no-cpu-baseline > $O/bench_cfg2_s16.json 2>/dev/null
bash tools/all_workloads.sh > $O/all_workloads.log 2>&1
# k_seg: same-box A/B against k_up2, the timing-only ablations, where a wave's cycles go, the ratio sweep
bash tools/ab_env.sh "-;CLOWNRESAMPLER_AMD_NO_SEG=1;-;CLOWNRESAMPLER_AMD_NO_SEG=1;CLOWNRESAMPLER_AMD_SEG_TILE=64" cfg3 > $O/kseg_ab.log 2>&1
for f in 1 2 3; do CLOWNRESAMPLER_AMD_SEG_FORM=$f python3 bench.py --workload cfg3 --no-check --no-cpu-baseline --no-host-paths --no-n1-reference 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('cfg3, k_seg timing-only form $f (1: no row loads, 2: no stores, 3: neither): %7.1f us' % (j['ms_per_step']*1e3))
"; done >> $O/kseg_ab.log 2>&1
python3 tools/kseg_phases.py cfg3 >> $O/kseg_ab.log 2>&1
timeout 900 python3 tools/seg_ratio_sweep.py > $O/seg_ratio_sweep.log 2>&1
for n in 2 8; do python3 bench.py --gpus $n > $O/bench_n${n}_sharedgpu_gloo.json 2> $O/bench_n$n.err; done
python3 tools/pinned_paths.py > $O/host_paths_pinned.log 2>&1
(python3 tools/channel_table.py 3
 python3 tools/channel_table.py 8 44100:48000 48000:44100 8000:96000) > $O/channel_table.log 2>&1
CRA_PREFLIGHT_DRY=1 timeout 2400 bash tools/multi_gpu_preflight.sh $O/preflight > $O/preflight_stdout.log 2>&1
rm -f $O/preflight/*.err
ls -la $O
tail -3 $O/gpu_tests.log
for w in cfg2 cfg2_steps20 cfg3 cfg4 cfg5 hq48 hq44 dn8; do python3 - <<PY
import json
l=json.loads([x for x in open("$O/bench_$w.json") if x.startswith("{")][0])
print("$w", l["roofline"]["kernel"], "%.1f us" % (l["ms_per_step"]*1e3), "median %.1f" % l["launch_us"]["median"], "frac %.3f" % l["roofline"]["frac"], "parity", l.get("parity_full_stream"), "traffic", l["roofline"].get("traffic"), "valu", (l.get("roofline_valu") or {}).get("frac"))
PY
done
