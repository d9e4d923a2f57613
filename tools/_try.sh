#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
python tools/channel_table.py 8 44100:48000 48000:44100 channels=6,7,8 2>&1 | tail -7
CLOWNRESAMPLER_AMD_VARIANT=14 python tools/channel_table.py 8 44100:48000 48000:44100 channels=6,7,8 2>&1 | tail -6
