#!/bin/bash
O=gpurun_out/try5; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?"; tail -3 $O/tests.log
for w in hq48 hq44 dn8 dn31 dn21 dn96 hq48c6; do echo "== $w"; python tools/sweep_variants.py --workload $w --rounds 5 --steps 20 --variants 30 2>&1 | tail -2; done
