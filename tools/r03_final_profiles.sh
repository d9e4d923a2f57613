#!/bin/bash
# Round 3 evidence, one box, ONE library build (the PMC summaries are stamped with its source id and bench.py quotes them only
# for that build): bench lines, rocprofv3 kernel stats and PMC passes for the BASELINE configurations and the long windows of
# VERDICT r2 item 1, every workload of bench.py's table, the N>1 validation runs, the channel tables, the host paths.
#   gpurun -- 'bash tools/r03_final_profiles.sh'      then copy gpurun_out/r03final/* into profiles/ under r03_ names
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03final
rm -rf $O; mkdir -p $O
for w in cfg2 cfg3 cfg4; do
  python3 $R/bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -- python3 $R/bench.py --workload $w --no-cpu-baseline --no-host-paths > $O/trace_$w.log 2>&1
  f=$(find $O/trace_$w -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$w.csv
  t=$(find $O/trace_$w -name "*kernel_trace.csv" | head -1); head -6 "$t" > $O/kernel_trace_head_$w.csv
  rm -rf $O/trace_$w
done
for w in cfg2 cfg3 cfg4 dn8 hq44 hq48 dn6x dn6xm; do
  bash $R/tools/pmc_passes.sh $O/pmc_$w --workload $w > $O/pmc_$w.log 2>&1
  cp $O/pmc_$w/pmc_summary.txt $O/pmc_summary_$w.txt
  rm -rf $O/pmc_$w
done
cd $R
# (second bench line of the BASELINE configurations, now that the stamped summaries exist: `traffic` and `roofline_valu` filled in)
mkdir -p $O/profiles_stage
for w in cfg2 cfg3 cfg4 dn8 hq44 hq48 dn6x dn6xm; do cp $O/pmc_summary_$w.txt profiles/r03_${w}_pmc_summary.txt; done
for w in cfg2 cfg3 cfg4; do python3 bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; done
for w in cfg5 dn8 hq44 hq48 dn6x dn6xm; do python3 bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>/dev/null; done
python3 bench.py --workload cfg2 --s16 --no-cpu-baseline > $O/bench_cfg2_s16.json 2>/dev/null
bash tools/all_workloads.sh > $O/all_workloads.log 2>&1
for n in 2 8; do python3 bench.py --gpus $n > $O/bench_n${n}_sharedgpu_gloo.json 2> $O/bench_n$n.err; done
(python3 tools/channel_table.py 3
 python3 tools/channel_table.py 3 48000:24000 48000:16000 48000:12000 48000:8000 44100:8000 channels=1,2,3,4,5,6,7,8 samples=105840000
 python3 tools/channel_table.py 8 44100:48000 48000:44100 8000:96000
 python3 tools/channel_table.py 5 44100:48000 48000:44100 channels=1,2,6,8
 python3 tools/channel_table.py 3 96000:48000 96000:44100 96000:32000 24000:48000 16000:48000 8000:48000 48000:32000 channels=1,2
 python3 tools/channel_table.py 8 96000:48000 96000:32000 24000:48000 12000:48000 48000:32000 channels=1,2
 python3 tools/channel_table.py 5 96000:48000 96000:32000 96000:24000 24000:48000 12000:48000 48000:32000 channels=1,2) > $O/channel_table.log 2>&1
python3 tools/channel_table.py 3 44100:16000 44100:22050 48000:16000 48000:8000 44100:11025 48000:22050 48000:24000 96000:16000 32000:44100 32000:48000 22050:44100 22050:48000 16000:44100 16000:48000 11025:44100 8000:48000 8000:16000 24000:44100 44100:32000 48000:32000 88200:48000 channels=1,2 > $O/common_ratios.log 2>&1
python3 tools/size_sweep.py > $O/size_sweep.log 2>&1
python3 tools/host_path_rate.py > $O/host_paths.log 2>&1
python3 tools/plan_create_rate.py > $O/plan_create.log 2>&1
ls -la $O
for w in cfg2 cfg3 cfg4 cfg5 dn8 hq44 hq48 dn6x dn6xm; do python3 - <<PY
import json
l=json.loads([x for x in open("$O/bench_$w.json") if x.startswith("{")][0])
print("$w", l["roofline"]["kernel"], "%.1f us" % (l["ms_per_step"]*1e3), "frac %.3f" % l["roofline"]["frac"], "traffic", l["roofline"].get("traffic"), "valu", (l.get("roofline_valu") or {}).get("frac"))
PY
done
head -3 $O/kernel_stats_cfg2.csv | cut -c1-200
