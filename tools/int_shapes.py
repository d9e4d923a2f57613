#!/usr/bin/env python3
"""The constants a k_int instance for a PERIODIC ratio is compiled for (cr_inst_int*.hip): host-only, no GPU.
   usage: int_shapes.py [radius] in:out[:lowpass] ...      e.g.  int_shapes.py 3 48000:32000 24000:48000 12000:48000"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import clownresampler_amd as cr

args = sys.argv[1:]
radius = int(args.pop(0)) if args and ":" not in args[0] else 3
api = cr.load(radius)
pre = api.precomputed()
for a in args:
    r = [int(x) for x in a.split(":")]
    if len(r) == 2:
        r.append(min(r))
    st = api.LowLevel_State()
    assert api.LowLevel_Init(st, 1, *r)
    sh = api.PeriodicShape(st.lowest_level, pre, st.increment)
    if sh is None:
        print("%s: increment %d is not periodic within 4 frames (or the rows do not fit)" % (a, st.increment))
        continue
    tt = sh["slots"]
    print("%s: increment %d, %d input frames per %d output frames, %d slots, starts %s" % (a, st.increment, sh["ratio"], sh["period"], tt, sh["starts"]))
    offs = sum(s << (8 * p) for p, s in enumerate(sh["starts"]))
    print("    make_per<CH, %d, %d, 0x%Xu, %d, K, 0x%Xull, 0x%Xull, 0x%Xull>()" % (sh["ratio"], sh["period"], offs, tt, sh["negmask"], sh["safemask"], sh["zeromask"]))
    for p in range(sh["period"]):
        print("    phase %d: signs %s" % (p, "".join("0" if (sh["zeromask"] >> (p * tt + s)) & 1 else ("-" if (sh["negmask"] >> (p * tt + s)) & 1 else "+") for s in range(tt))))
