#!/bin/bash
O=gpurun_out/${1:-r02d}
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -k "input_stationary or kernel_variants or cfg3 or clamped" > $O/kup_tests.log 2>&1; tail -5 $O/kup_tests.log
python tools/sweep_variants.py --workload cfg3 --variants 26,27 > $O/sweep_cfg3.log 2>&1; tail -4 $O/sweep_cfg3.log
python tools/sweep_variants.py --workload up55 --variants 13,26,27 > $O/sweep_up55.log 2>&1; tail -5 $O/sweep_up55.log
