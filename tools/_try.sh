#!/bin/bash
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for r in 48000:16000 48000:19200 44100:32000; do for m in 18 999; do echo "== 8 lobes $r min slots $m"; CLOWNRESAMPLER_AMD_RT_WAVE2_MIN_SLOTS=$m python tools/channel_table.py 8 $r channels=2,3,4,5,6,7 2>&1 | grep -v "^radius\|^ch |amdgpu"; done; done
for r in 48000:15000; do for m in 18 999; do echo "== 3 lobes $r min slots $m"; CLOWNRESAMPLER_AMD_RT_WAVE2_MIN_SLOTS=$m python tools/channel_table.py 3 $r channels=2,3,4,5,6,7 2>&1 | grep -v "^radius\|^ch |amdgpu"; done; done
