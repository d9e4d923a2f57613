#!/bin/bash
# Round 6 evidence, one box, ONE library build (the PMC summaries are stamped with its source id; bench.py quotes them only for that build):
# bench lines + rocprofv3 kernel traces for the BASELINE configurations with the mean over the TIMED dispatches taken from the trace
# (tools/trace_timed_mean.py), PMC passes for them, the long-window shapes' lines, the host paths; then the GPU suite in a loop of fresh processes.
#   gpurun --timeout 3300 -- 'bash tools/r06_final_profiles.sh'     then     python tools/r06_install_evidence.py
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06final
rm -rf $O; mkdir -p $O
for w in cfg2 cfg3 cfg4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -- python3 $R/bench.py --workload $w --no-cpu-baseline --no-host-paths --no-n1-reference > $O/trace_$w.json 2> $O/trace_$w.err
  f=$(find $O/trace_$w -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$w.csv
  t=$(find $O/trace_$w -name "*kernel_trace.csv" | head -1); head -6 "$t" > $O/kernel_trace_head_$w.csv
  python3 $R/tools/trace_timed_mean.py "$t" $O/trace_$w.json >> $O/trace_timed_means.log 2>&1
  rm -rf $O/trace_$w $O/trace_$w.err
done
cat $O/trace_timed_means.log
for w in cfg2 cfg3 cfg4; do
  bash $R/tools/pmc_passes.sh $O/pmc_$w --workload $w > $O/pmc_$w.log 2>&1
  cp $O/pmc_$w/pmc_summary.txt $O/pmc_summary_$w.txt
  rm -rf $O/pmc_$w $O/pmc_$w.log
done
cd $R
for w in cfg2 cfg3 cfg4; do cp $O/pmc_summary_$w.txt profiles/r06_${w}_pmc_summary.txt; done
python3 bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python3 bench.py --steps 20 --warmup 3 > $O/bench_cfg2_steps20.json 2>/dev/null     # as the driver runs it
for w in cfg3 cfg4; do python3 bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; done
for w in cfg5 hq48 hq44 dn8; do python3 bench.py --workload $w --no-cpu-baseline --no-host-paths > $O/bench_$w.json 2>/dev/null; done
python3 bench.py --workload cfg2 --s16 --no-cpu-baseline > $O/bench_cfg2_s16.json 2>/dev/null
bash tools/ab_env.sh "-;CLOWNRESAMPLER_AMD_SEG_XCD_RUN=0;-;CLOWNRESAMPLER_AMD_SEG_XCD_RUN=0;CLOWNRESAMPLER_AMD_NO_SEG=1" cfg3 > $O/kseg_ab.log 2>&1
python3 bench.py --gpus 2 > $O/bench_n2_sharedgpu_gloo.json 2> $O/bench_n2.err
python3 tools/host_path_rate.py > $O/host_paths.log 2>&1
for w in cfg2 cfg2_steps20 cfg3 cfg4 cfg5 hq48 hq44 dn8 cfg2_s16; do python3 - <<PY
import json
l=json.loads([x for x in open("$O/bench_$w.json") if x.startswith("{")][0])
print("$w", l["roofline"]["kernel"], "%.1f us" % (l["ms_per_step"]*1e3), "median %.1f" % l["launch_us"]["median"], "frac %.3f" % l["roofline"]["frac"], "parity", l.get("parity_full_stream"), "traffic", l["roofline"].get("traffic"), "value %.0f" % l["value"])
PY
done | tee $O/lines.log
rm -f $O/*.err
bash tools/experiments/r06/loop_suite.sh d ${LOOP_RUNS:-12}
