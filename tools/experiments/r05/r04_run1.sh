#!/bin/bash
# round 4, first GPU call: new tests, bench lines (N = 1, N = 2 validation), page-locked host paths, PMC summaries of the mono workloads
set -u
O=gpurun_out/r04a; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/gpu_tests.log 2>&1
timeout 300 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_n2.json 2> $O/bench_n2.err
timeout 900 python tools/pinned_paths.py > $O/pinned_paths.log 2>&1
for w in mono dn1 hq48m hq44m dn8m; do
  timeout 600 bash tools/pmc_passes.sh $O/pmc_$w --workload $w > $O/pmc_$w.log 2>&1
  cp $O/pmc_$w/pmc_summary.txt $O/r04_${w}_before_pmc_summary.txt 2>/dev/null
  rm -rf $O/pmc_$w/pmc*/
done
timeout 900 bash tools/all_workloads.sh > $O/all_workloads.log 2>&1
echo done
