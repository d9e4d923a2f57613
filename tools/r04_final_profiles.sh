#!/bin/bash
# Round 4 evidence, one box, ONE library build (the PMC summaries are stamped with its source id; bench.py quotes them only for that
# build): the GPU test suite, bench lines + rocprofv3 kernel stats + PMC passes for the BASELINE configurations, PMC passes for the
# mono workloads, every workload of bench.py's table, the N > 1 validation runs on the one GPU, the channel tables, the host paths.
#   gpurun --timeout 3000 -- 'bash tools/r04_final_profiles.sh'     then     python tools/r04_install_evidence.py
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04final
rm -rf $O; mkdir -p $O
( cd $R; timeout 1500 python3 -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/gpu_tests.log 2>&1
for w in cfg2 cfg3 cfg4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -- python3 $R/bench.py --workload $w --no-cpu-baseline --no-host-paths --no-n1-reference > $O/trace_$w.log 2>&1
  f=$(find $O/trace_$w -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$w.csv
  t=$(find $O/trace_$w -name "*kernel_trace.csv" | head -1); head -6 "$t" > $O/kernel_trace_head_$w.csv
  rm -rf $O/trace_$w $O/trace_$w.log
done
for w in cfg2 cfg3 cfg4 mono dn1 hq48m hq44m dn8m; do
  bash $R/tools/pmc_passes.sh $O/pmc_$w --workload $w > $O/pmc_$w.log 2>&1
  cp $O/pmc_$w/pmc_summary.txt $O/pmc_summary_$w.txt
  rm -rf $O/pmc_$w $O/pmc_$w.log
done
cd $R
# the bench lines, now that the stamped summaries exist (`traffic` and `roofline_valu` filled in for the BASELINE configurations)
for w in cfg2 cfg3 cfg4; do cp $O/pmc_summary_$w.txt profiles/r04_${w}_pmc_summary.txt; done
python3 bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
for w in cfg3 cfg4; do python3 bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; done
for w in cfg5 hq48; do python3 bench.py --workload $w --no-cpu-baseline --no-host-paths > $O/bench_$w.json 2>/dev/null; done
python3 bench.py --workload cfg2 --s16 --no-cpu-baseline > $O/bench_cfg2_s16.json 2>/dev/null
bash tools/all_workloads.sh > $O/all_workloads.log 2>&1
for n in 2 8; do python3 bench.py --gpus $n > $O/bench_n${n}_sharedgpu_gloo.json 2> $O/bench_n$n.err; done
python3 tools/pinned_paths.py > $O/host_paths_pinned.log 2>&1
# k_up2 on cfg 3: the default (27, FP32 chain), the integer chain (26), and the timing-only ablations (1008 clock stamps, 1009 no stores, 1010 no frames)
python3 tools/sweep_variants.py --workload cfg3 --variants 27,26,1008,1009,1010 > $O/kup2_ablations.log 2>&1
(python3 tools/channel_table.py 3
 python3 tools/channel_table.py 8 44100:48000 48000:44100 8000:96000) > $O/channel_table.log 2>&1
ls -la $O
tail -3 $O/gpu_tests.log
for w in cfg2 cfg3 cfg4 cfg5 hq48; do python3 - <<PY
import json
l=json.loads([x for x in open("$O/bench_$w.json") if x.startswith("{")][0])
print("$w", l["roofline"]["kernel"], "%.1f us" % (l["ms_per_step"]*1e3), "frac %.3f" % l["roofline"]["frac"], "parity", l.get("parity_full_stream"), "traffic", l["roofline"].get("traffic"), "valu", (l.get("roofline_valu") or {}).get("frac"))
PY
done
