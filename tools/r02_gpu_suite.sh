#!/bin/bash
# the whole -m gpu suite + a default bench line
O=gpurun_out/${1:-r02b}
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
echo "pytest rc=$?"; tail -15 $O/gpu_tests.log
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
head -c 600 $O/bench_n1.json; echo
