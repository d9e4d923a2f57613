#!/bin/bash
CLOWNRESAMPLER_AMD_LIBRARY=$PWD/tools/ab/libSA.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
bash tools/ab.sh 27 "A SA" cfg3
bash tools/ab.sh 30 "A SA" hq48 dn31
bash tools/ab.sh 13 "A SA" cfg2
