#!/bin/bash
# channel table under two library builds (tools/ab/lib<NAME>.so), one box: ab_table.sh "<lib names>" <channel_table.py args...>
LIBS=$1; shift
for L in $LIBS; do echo "== lib$L"; CLOWNRESAMPLER_AMD_LIBRARY=$PWD/tools/ab/lib$L.so python tools/channel_table.py "$@" 2>&1 | grep -E "^ *[0-9]+ \|"; done
