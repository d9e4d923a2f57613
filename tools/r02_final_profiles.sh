#!/bin/bash
# Round 2 evidence, one box: bench lines, rocprofv3 kernel stats and PMC passes for the BASELINE configurations (+ the 8-lobe
# 44.1 -> 48 kHz shape k_wave2 serves), the N>1 validation runs, the channel table and the size sweep.
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02final
rm -rf $O; mkdir -p $O
for w in cfg2 cfg3 cfg4 hq48; do
  python3 $R/bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -- python3 $R/bench.py --workload $w --no-cpu-baseline > $O/trace_$w.log 2>&1
  f=$(find $O/trace_$w -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$w.csv
  t=$(find $O/trace_$w -name "*kernel_trace.csv" | head -1); head -6 "$t" > $O/kernel_trace_head_$w.csv
  rm -rf $O/trace_$w
  bash $R/tools/pmc_passes.sh $O/pmc_$w --workload $w > $O/pmc_$w.log 2>&1
  cp $O/pmc_$w/pmc_summary.txt $O/pmc_summary_$w.txt
  rm -rf $O/pmc_$w
done
cd $R
python3 bench.py --workload cfg5 --no-cpu-baseline > $O/bench_cfg5.json 2>/dev/null
for w in hq44 dn8 dn21 dn96 dn31; do python3 bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>/dev/null; done
python3 bench.py --workload cfg2 --s16 --no-cpu-baseline > $O/bench_cfg2_s16.json 2>/dev/null
for n in 2 8; do python3 bench.py --gpus $n > $O/bench_n${n}_sharedgpu_gloo.json 2> $O/bench_n$n.err; done
(python3 tools/channel_table.py 3; python3 tools/channel_table.py 8 44100:48000 48000:44100; python3 tools/channel_table.py 3 96000:48000 96000:44100 96000:32000 channels=1,2) > $O/channel_table.log 2>&1
# the specialised k_poly the 17-slot k_wave2 entries fall back to where a ratio's rows do not fit, against the run-time-slot instance
(CLOWNRESAMPLER_AMD_VARIANT=13 python3 tools/channel_table.py 8 48000:44100 channels=1,2,3,4,5; CLOWNRESAMPLER_AMD_NO_SPECIAL=1 python3 tools/channel_table.py 8 48000:44100 channels=1,2,3,4,5) > $O/fallback_17slot.log 2>&1
python3 tools/size_sweep.py > $O/size_sweep.log 2>&1
ls -la $O
for w in cfg2 cfg3 cfg4 hq48 cfg5 hq44 dn8 dn21 dn96 dn31; do python3 - <<PY
import json
l=json.loads([x for x in open("$O/bench_$w.json") if x.startswith("{")][0])
print("$w", l["roofline"]["kernel"], "%.1f us" % (l["ms_per_step"]*1e3), "frac %.3f" % l["roofline"]["frac"], "valu", (l.get("roofline_valu") or {}).get("frac"))
PY
done
head -3 $O/kernel_stats_cfg2.csv | cut -c1-200
