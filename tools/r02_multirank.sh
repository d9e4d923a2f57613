#!/bin/bash
# Round 2: the N>1 bench path on the pool's 1-GPU box (ranks share GPU 0, gloo: validation of everything but RCCL itself).
O=gpurun_out/${1:-r02a}
mkdir -p $O
python -m pytest tests/test_gpu_ranks.py -x -q > $O/ranks_test.log 2>&1
tail -3 $O/ranks_test.log
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --one-event-pair --no-cpu-baseline > $O/bench_n1_onepair.json 2> $O/bench_n1_onepair.err
python bench.py --workload cfg5 --no-cpu-baseline > $O/bench_n1_cfg5.json 2> $O/bench_n1_cfg5.err
for n in 2 4 8; do
  python bench.py --gpus $n > $O/bench_n$n.json 2> $O/bench_n$n.err
  echo "n=$n rc=$?"
done
python bench.py --gpus 2 --scaling weak --workload cfg2 > $O/bench_n2_weak.json 2> $O/bench_n2_weak.err
head -c 1500 $O/bench_n1.json; echo; head -c 600 $O/bench_n1_onepair.json; echo
for n in 2 8; do head -c 2500 $O/bench_n$n.json; echo; tail -5 $O/bench_n$n.err; done
