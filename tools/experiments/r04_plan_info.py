#!/usr/bin/env python3
"""prints the plan (kernel, geometry, LDS bytes, grid cap) bench.py's workloads get"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, clownresampler_amd as cr
import torch; torch.cuda.init()
for w in sys.argv[1:]:
    radius, ch, rates, frames = bench.WORKLOADS[w]
    api = cr.load(radius)
    st = api.LowLevel_State(); api.LowLevel_Init(st, ch, *rates)
    info = api.PlanGetInfo(api.PlanCreate(st, api.precomputed()))
    print("%-8s" % w, info.asdict())
