#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05run54; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
bash $R/tools/pmc_passes.sh $O/pmc_cfg2_s16 --workload cfg2 --s16 > $O/pmc_cfg2_s16.log 2>&1; cp $O/pmc_cfg2_s16/pmc_summary.txt $O/pmc_summary_cfg2_s16.txt; rm -rf $O/pmc_cfg2_s16
for w in dn1 mono dn2; do
  bash $R/tools/pmc_passes.sh $O/pmc_$w --workload $w > $O/pmc_$w.log 2>&1; cp $O/pmc_$w/pmc_summary.txt $O/pmc_summary_$w.txt; rm -rf $O/pmc_$w
done
ls $O; head -30 $O/pmc_summary_cfg2_s16.txt
