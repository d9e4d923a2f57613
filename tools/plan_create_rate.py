#!/usr/bin/env python3
"""Host cost of creating plans that differ only in their increment (the variable-rate path makes one per distinct ratio)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import clownresampler_amd as cr
api = cr.load(3); pre = api.precomputed()
api.SetPlanCacheLimit(64) if hasattr(api, "SetPlanCacheLimit") else None
for ch in (2, 1):
    st = api.LowLevel_State(); api.LowLevel_Init(st, ch, 44100, 48000, 44100); api.PlanCreate(st, pre)
    t = []
    for k in range(1, 301):
        st = api.LowLevel_State(); api.LowLevel_Init(st, ch, 44100, 48000 + k, 44100)
        t0 = time.perf_counter(); api.PlanCreate(st, pre); t.append(time.perf_counter() - t0)
    t.sort()
    print("ch %d: PlanCreate of a sibling plan: median %.1f us, min %.1f, p90 %.1f" % (ch, t[150] * 1e6, t[0] * 1e6, t[270] * 1e6))
