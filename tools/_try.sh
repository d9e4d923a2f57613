#!/bin/bash
CLOWNRESAMPLER_AMD_LIBRARY=$PWD/tools/ab/libB.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
bash tools/ab_bench.sh "A B" cfg2 cfg4 cfg3 cfg5
