#!/bin/bash
O=gpurun_out/try4; mkdir -p $O
python tools/channel_table.py 3 96000:48000 96000:44100 96000:32000 44100:8000 channels=2 2>&1 | tail -5
python tools/channel_table.py 8 44100:48000 48000:44100 channels=1,2,3,4,5,6 2>&1 | tail -13
for w in dn8 dn31 hq48; do echo "== $w"; python tools/sweep_variants.py --workload $w --rounds 5 --steps 20 --variants 13,30 2>&1 | tail -3; done
python tools/size_sweep.py 2>&1 | tail -12
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?"; tail -3 $O/tests.log
