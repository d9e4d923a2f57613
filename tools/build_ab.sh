#!/bin/bash
# Builds the library as it is at a git revision into tools/ab/lib<NAME>.so (for tools/ab_bench.sh): build_ab.sh <NAME> [<rev>]
# The working tree's own build is tools/ab/libnew.so (a copy of the in-tree library).
set -e
NAME=$1; REV=${2:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/tools/ab
if [ "$NAME" = "new" ]; then cp $ROOT/clownresampler_amd/libclownresampler_amd.so $ROOT/tools/ab/libnew.so; exit 0; fi
T=$(mktemp -d /tmp/crab.XXXXXX)
git -C $ROOT archive $REV clownresampler_amd/csrc include tools/cr_resample.c tools/cr_multi.c tools/cb_store.c 2>/dev/null | tar -x -C $T || git -C $ROOT archive $REV clownresampler_amd/csrc include tools/cr_resample.c tools/cr_multi.c | tar -x -C $T
make -s -j8 -C $T/clownresampler_amd/csrc $T/clownresampler_amd/libclownresampler_amd.so > /dev/null
cp $T/clownresampler_amd/libclownresampler_amd.so $ROOT/tools/ab/lib$NAME.so
rm -rf $T
echo built tools/ab/lib$NAME.so from $REV
