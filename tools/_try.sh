#!/bin/bash
O=gpurun_out/try7; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?"; tail -3 $O/tests.log
bash tools/ab.sh 30 "A ST" hq48 dn8 dn31 dn21 hq44 hq48c6
