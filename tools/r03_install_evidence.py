#!/usr/bin/env python3
"""Copies gpurun_out/r03final/* (tools/r03_final_profiles.sh) into profiles/ under r03_ names and rewrites the tables of DESIGN.md
section 5 that quote them.  usage: r03_install_evidence.py <label for the lines being replaced, e.g. box4>"""
import sys, os, re, json, shutil, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(ROOT, "gpurun_out", "r03final"), os.path.join(ROOT, "profiles")
old_label = sys.argv[1]
for w in ("cfg2", "cfg3", "cfg4", "cfg5", "hq48"):
    shutil.copy(os.path.join(P, "r03_bench_%s.json" % w), os.path.join(P, "r03_%s_bench_%s.json" % (old_label, w)))
for w in ("cfg2", "cfg3", "cfg4", "dn8", "hq44", "hq48", "dn6x", "dn6xm"):
    shutil.copy(os.path.join(O, "pmc_summary_%s.txt" % w), os.path.join(P, "r03_%s_pmc_summary.txt" % w))
for w in ("cfg2", "cfg3", "cfg4"):
    shutil.copy(os.path.join(O, "kernel_stats_%s.csv" % w), os.path.join(P, "r03_kernel_stats_%s.csv" % w))
    shutil.copy(os.path.join(O, "kernel_trace_head_%s.csv" % w), os.path.join(P, "r03_kernel_trace_head_%s.csv" % w))
for w in ("cfg2", "cfg3", "cfg4", "cfg5", "dn8", "hq44", "hq48", "dn6x", "dn6xm", "cfg2_s16", "n2_sharedgpu_gloo", "n8_sharedgpu_gloo"):
    shutil.copy(os.path.join(O, "bench_%s.json" % w), os.path.join(P, "r03_bench_%s.json" % w))
for f in ("all_workloads", "channel_table", "common_ratios", "size_sweep", "host_paths", "plan_create"):
    text = [l for l in open(os.path.join(O, f + ".log")) if "amdgpu.ids" not in l]
    if f == "common_ratios":
        text.insert(0, "# tools/channel_table.py 3 <common conversions> channels=1,2 - one MI355X box, the final build of round 3 (kernel 5 = k_int, 4 = k_wave2, 1 = k_poly); round 2: profiles/r02_common_ratios.log\n")
    open(os.path.join(P, "r03_%s.log" % f), "w").writelines(text)
shutil.copy(os.path.join(ROOT, "gpurun_out", "r03_gpu_tests.log"), os.path.join(P, "r03_gpu_tests.log"))


def line(w):
    return json.loads([x for x in open(os.path.join(P, "r03_bench_%s.json" % w)) if x.startswith("{")][0])


def stats(w):
    f = open(os.path.join(P, "r03_kernel_stats_%s.csv" % w)).read().split("\n")[1].split(",")
    return int(f[-7]), float(f[-5]) / 1e3


wl = {}
for l in open(os.path.join(P, "r03_all_workloads.log")):
    f = l.split()
    if len(f) > 8 and f[3] == "us":
        wl[f[0]] = (f[1], float(f[2]), float(f[4]), float(f[7]))
out = {}
for w in ("cfg2", "cfg3", "cfg4", "cfg5", "dn8", "hq44", "hq48", "dn6x", "dn6xm", "cfg2_s16"):
    l = line(w); r = l["roofline"]; v = l.get("roofline_valu") or {}
    out[w] = dict(v=l["value"], us=l["ms_per_step"] * 1e3, gb=r["achieved"], frac=r["frac"], valu=v.get("frac"),
                  tr=(r["traffic"] / r["algorithmic_bytes_per_launch"]) if r.get("traffic") else None)
c2 = line("cfg2")
n2, n8 = line("n2_sharedgpu_gloo")["value"], line("n8_sharedgpu_gloo")["value"]
k = lambda x: "{:,.0f}".format(round(x, -2))


def row(name, kern, w, bold=True, rocprof=None, valubold=False):
    o = out[w]
    us = "%.1f" % o["us"] + ((" (%.1f over %s calls incl. warm-up)" % (rocprof[1], "{:,}".format(rocprof[0]))) if rocprof and w == "cfg2" else (" (%.1f)" % rocprof[1] if rocprof else ""))
    fr = ("**%.3f**" if bold else "%.3f") % o["frac"]
    va = "" if o["valu"] is None else (("**%.2f**" if valubold else "%.2f") % o["valu"])
    tr = "" if o["tr"] is None else ("%.3f" % o["tr"] if o["tr"] < 1.02 else "%.2f" % o["tr"])
    return "| %s | %s | %s | %s | %s | %s | %s | %s |" % (name, kern, k(o["v"]), us, "{:,.0f}".format(o["gb"]), fr, va, tr)


def multi(name, kern, ws, bold=True):
    f = " / ".join("%.3f" % wl[w][3] for w in ws)
    return "| %s | %s | %s | %s | | %s | | |" % (name, kern, " / ".join(k(wl[w][2]) for w in ws), " / ".join("%.1f" % wl[w][1] for w in ws), ("**%s**" % f) if bold else f)


rows = [
    row("cfg 2 stereo 44.1→48 kHz, 3 lobes, 10 min (default)", "`k_poly<2,5>` mov-armed chain", "cfg2", rocprof=stats("cfg2")),
    row("cfg 5 same, 1 hour on one GPU", "`k_poly<2,5>`", "cfg5"),
    row("cfg 4 8-ch 48→44.1 kHz", "`k_poly<8,6>` any-sign chain (round 2: SDWA, 247.7 µs, 0.660)", "cfg4", rocprof=stats("cfg4")),
    row("cfg 3 stereo 8→96 kHz, 8 lobes", "`k_up2<2,15>`", "cfg3", rocprof=stats("cfg3"), valubold=True),
    row("stereo 44.1→48 kHz, 8 lobes (`hq48`)", "`k_wave2<2,15>`", "hq48", bold=False),
    row("stereo 48→44.1 kHz, 8 lobes (`hq44`)", "`k_wave2<2,17>`", "hq44", bold=False),
    row("stereo 44.1→8 kHz, 33-slot windows (`dn8`)", "`k_wave2<2,33>`", "dn8", bold=False),
    row("stereo 48→8 kHz, 36-slot windows (`dn6x`; round 2: 130 µs, 0.295)", "**`k_int<2,36>`**", "dn6x"),
    row("mono 48→8 kHz (`dn6xm`; round 2: 152 µs, 0.25)", "**`k_int<1,36>`**", "dn6xm"),
    multi("stereo 2:1 / 3:1 / 4:1 (`dn21`, `dn31`, `dn4x`; round 2: `k_wave2` 0.49 / 0.49, `k_poly` 0.35)", "**`k_int<2,12/18/24>`**", ("dn21", "dn31", "dn4x")),
    multi("mono 2:1 (`dn21m`; round 2: 0.39)", "**`k_int<1,12>`**", ("dn21m",)),
    multi("4 / 6 / 8 channels 2:1 (`dn21c4/6/8`; before: 0.46 / 0.47 / 0.49)", "**`k_int<4/6/8,12>`**", ("dn21c4", "dn21c6", "dn21c8")),
    multi("stereo 3:2 (`dn32`, 48→32 kHz; round 2: 0.519)", "**`k_int<2,9>`**, two rows per launch", ("dn32",)),
    multi("stereo 48→44.1 kHz, 3 lobes (`dn2`; round 2: 0.559)", "`k_poly<2,6>` any-sign chain", ("dn2",), bold=False),
    multi("3 channels 48→44.1 kHz (`ch3`; round 2: 0.498)", "`k_poly<3,6>` any-sign chain", ("ch3",), bold=False),
    multi("8 lobes 48→44.1 kHz 6 channels (`hq44c6`; round 2: 0.294)", "`k_poly<6,17>` run-time slots, any-sign chain", ("hq44c6",), bold=False),
    row("cfg 2 with the opt-in int16 output (`--s16`)", "`k_poly<2,5>`", "cfg2_s16", bold=False),
    "| reference C path, 1 host thread (cfg 2, same box, `cpu_baseline`) | | ≈ %.0f | | | | | |" % c2["cpu_baseline"]["value"],
]
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
a = s.index("| cfg 2 stereo 44.1→48 kHz, 3 lobes, 10 min (default) | `k_poly<2,5>` mov-armed chain |")
b = s.index('\n("HBM traffic" = (2 × `FETCH_SIZE`')
s = s[:a] + "\n".join(rows) + "\n" + s[b:]
e = c2["end_to_end"]; cb = c2["callback_api"]
s = re.sub(r"`ClownResampler_LowLevel_ResampleBulk` over cfg 2 from pageable host memory [\d.]+ ms \([\d,]+ Msamples/s; first touch \d+ ms\), pinned [\d.]+\s+ms; the reference's callback signature \d+ ms \([\d,]+ Msamples/s;",
           "`ClownResampler_LowLevel_ResampleBulk` over cfg 2 from pageable host memory %.1f ms (%s Msamples/s; first touch %.0f ms), pinned %.1f\nms; the reference's callback signature %.0f ms (%s Msamples/s;" % (e["pageable_ms"], k(e["Msamples/s"]["pageable"]), e["pageable_first_ms"], e["pinned_ms"], cb["ms"], "{:,.0f}".format(cb["Msamples/s"])), s)
s = re.sub(r'N = 2 / 8 ranks sharing this one GPU over gloo \(validation only\): [\d,]+ / [\d,]+\s+"aggregate" Msamples/s\.',
           'N = 2 / 8 ranks sharing this one GPU over gloo (validation only): %s / %s\n"aggregate" Msamples/s.' % (k(n2), k(n8)), s)


def parse(path):
    d = {}; rad = None
    for l in open(path):
        m = re.match(r"radius (\d+)", l)
        if m:
            rad = int(m.group(1)); continue
        m = re.match(r" *(\d+) \| *(\d+) -> *(\d+) \| (\d+) +(\d+) +(\d+) \| +([\d.]+) \| +(\d+) \| +(\d+) \| ([\d.]+)", l)
        if m:
            d.setdefault((rad, int(m.group(2)), int(m.group(3))), {})[int(m.group(1))] = (int(m.group(4)), float(m.group(7)), float(m.group(10)))
    return d


t = parse(os.path.join(P, "r03_channel_table.log"))
names = {(3, 44100, 48000): "3 lobes 44100→48000 (5 taps)", (3, 48000, 44100): "3 lobes 48000→44100 (6 taps)", (3, 44100, 8000): "3 lobes 44100→8000 (33 taps)",
         (3, 8000, 44100): "3 lobes 8000→44100 (5 taps)", (8, 44100, 48000): "8 lobes 44100→48000 (15 taps)", (8, 48000, 44100): "8 lobes 48000→44100 (17 taps)",
         (8, 8000, 96000): "8 lobes 8000→96000 (15 taps)"}
trows = []
for key, n in names.items():
    cells = []
    for c in range(1, 17):
        cell = ("**%.2f**" if t[key][c][0] == 4 else "%.2f") % t[key][c][2]
        if key == (8, 8000, 96000) and c == 2:
            cell = "%.2f (`k_up2`)" % t[key][c][2]
        cells.append(cell)
    trows.append("| %s | %s |" % (n, " | ".join(cells)))
trows.append("| 3 lobes 2:1 / 3:1 / 4:1 / 6:1 (`k_int`) | " + " | ".join(" / ".join("%.2f" % t[(3, 48000, o)][c][2] for o in (24000, 16000, 12000, 8000)) for c in range(1, 9)) + " | | | | | | | | |")
ms = lambda key: " / ".join("%.2f" % v[2] for c, v in sorted(t[key].items()))
extra = """
Mono / stereo at further ratios (same file): 3:2 %s (`k_int`); 96 → 44.1 kHz %s (stereo: `k_wave2`); 24 → 48, 16 → 48, 8 → 48 kHz %s,
%s, %s; 8 lobes 2:1 %s, 3:1 %s, 1:2 %s, 1:4 %s, 3:2 %s (all `k_int`);
5 lobes 2:1 %s, 3:1 %s, 4:1 %s, 1:2 %s, 1:4 %s, 3:2 %s (`k_int`), 44.1 → 48 kHz
%s, 48 → 44.1 kHz %s (run-time-slot `k_poly`). """ % (ms((3, 48000, 32000)), ms((3, 96000, 44100)), ms((3, 24000, 48000)), ms((3, 16000, 48000)), ms((3, 8000, 48000)), ms((8, 96000, 48000)), ms((8, 96000, 32000)),
    ms((8, 24000, 48000)), ms((8, 12000, 48000)), ms((8, 48000, 32000)), ms((5, 96000, 48000)), ms((5, 96000, 32000)), ms((5, 96000, 24000)), ms((5, 24000, 48000)), ms((5, 12000, 48000)),
    ms((5, 48000, 32000)), " / ".join("%.2f" % t[(5, 44100, 48000)][c][2] for c in (1, 2)), " / ".join("%.2f" % t[(5, 48000, 44100)][c][2] for c in (1, 2)))
a = s.index("| 3 lobes 44100→48000 (5 taps) |")
b = s.index("The everyday conversions, mono and stereo, beside round 2's:")
s = s[:a] + "\n".join(trows) + "\n" + extra + s[b:]
open(p, "w").write(s)
print("installed; cfg2 %.1f us %.3f, cfg3 %.1f, cfg4 %.1f; source id %s" % (out["cfg2"]["us"], out["cfg2"]["frac"], out["cfg3"]["us"], out["cfg4"]["us"],
      open(os.path.join(P, "r03_cfg2_pmc_summary.txt")).readline().split()[1]))
