#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06/hostpath; mkdir -p $O
for piece in 1048576 0 1048576 0; do
  echo "== CLOWNRESAMPLER_AMD_PAGEABLE_PIECE=$piece (GPU_PINNED_MIN_XFER_SIZE unset)" >> $O/host_path.log
  env -u GPU_PINNED_MIN_XFER_SIZE CLOWNRESAMPLER_AMD_PAGEABLE_PIECE=$piece timeout 600 python tools/host_path_rate.py >> $O/host_path.log 2>&1
done
tail -40 $O/host_path.log
bash tools/experiments/r06/loop_suite.sh ${1:-b} ${2:-10}
