#!/bin/bash
# A/B timing of several builds of the library on ONE box (tools/ab/lib<NAME>.so), alternating processes.
#   bash tools/ab.sh "<variants>" "<lib names>" <workload> [<workload> ...]
V=$1; shift
LIBS=$1; shift
for w in "$@"; do
  for rep in 1 2; do
    for L in $LIBS; do
      r=$(CLOWNRESAMPLER_AMD_LIBRARY=$PWD/tools/ab/lib$L.so python tools/sweep_variants.py --workload $w --rounds 5 --steps 20 --variants $V 2>&1 | grep -E "^ *[0-9]+ " | awk '{printf "v%s %s/%s  ", $1, $(NF-4), $(NF-3)}')
      echo "$w lib$L: $r"
    done
  done
done
