import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import clownresampler_amd as cr
api = cr.load(3); pre = api.precomputed()
for ch in (1, 2, 4):
    for rates in [(44100, 48000), (24000, 48000), (16000, 48000), (12000, 48000), (8000, 44100), (8000, 48000), (12000, 96000), (8000, 80000), (8000, 96000), (12000, 192000), (11025, 48000), (22050, 48000), (32000, 48000), (44100, 96000), (44100, 192000), (48000, 96000), (9000, 192000), (7500, 96000)]:
        st = api.LowLevel_State(); assert api.LowLevel_Init(st, ch, rates[0], rates[1], rates[0])
        print("ch", ch, rates, file=sys.stderr, flush=True)
        api.PlanCreate(st, pre)
