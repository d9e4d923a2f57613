#!/bin/bash
# k_seg: tiles in runs of consecutive tiles per XCD (crhip_seg_launch.xcd_run) against one by one - launch time (same box, alternating) and HBM reads.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06/kseg_xcd; mkdir -p $O
for rep in 1 2 3; do
  bash tools/ab_env.sh "CLOWNRESAMPLER_AMD_SEG_XCD_RUN=0;CLOWNRESAMPLER_AMD_SEG_XCD_RUN=4;CLOWNRESAMPLER_AMD_SEG_XCD_RUN=8;CLOWNRESAMPLER_AMD_SEG_XCD_RUN=16;CLOWNRESAMPLER_AMD_SEG_XCD_RUN=32;CLOWNRESAMPLER_AMD_SEG_XCD_RUN=64;CLOWNRESAMPLER_AMD_SEG_XCD_RUN=128" cfg3 2>&1 | tee -a $O/ab.log
done
for g in 0 16; do
  CLOWNRESAMPLER_AMD_SEG_XCD_RUN=$g bash tools/pmc_passes.sh $O/pmc_run$g --workload cfg3 > $O/pmc_run$g.log 2>&1
  echo "== xcd_run $g" | tee -a $O/pmc.log; grep -a "FETCH_SIZE\|WRITE_SIZE\|TCC_HIT\|TCC_MISS\|TCC_EA0_RDREQ" $O/pmc_run$g/pmc_summary.txt | tee -a $O/pmc.log
done
