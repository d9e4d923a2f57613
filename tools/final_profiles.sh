set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final2
rm -rf $O; mkdir -p $O
for w in cfg2 cfg3 cfg4; do
  python3 $R/bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -- python3 $R/bench.py --workload $w --no-cpu-baseline > $O/trace_$w.log 2>&1
  f=$(find $O/trace_$w -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$w.csv
  t=$(find $O/trace_$w -name "*kernel_trace.csv" | head -1); head -6 "$t" > $O/kernel_trace_head_$w.csv
  rm -rf $O/trace_$w
  bash $R/tools/pmc_passes.sh $O/pmc_$w --workload $w > $O/pmc_$w.log 2>&1
  cp $O/pmc_$w/pmc_summary.txt $O/pmc_summary_$w.txt
  rm -rf $O/pmc_$w
done
python3 $R/bench.py --workload cfg5 --no-cpu-baseline > $O/bench_cfg5.json 2>/dev/null
python3 $R/bench.py --workload cfg2 --s16 --no-cpu-baseline > $O/bench_cfg2_s16.json 2>/dev/null
python3 $R/bench.py --workload mono --no-cpu-baseline > $O/bench_mono.json 2>/dev/null
python3 $R/bench.py --workload ch16 --no-cpu-baseline > $O/bench_ch16.json 2>/dev/null
ls -la $O
head -3 $O/kernel_stats_cfg2.csv | cut -c1-200
