#!/usr/bin/env python3
"""Copies gpurun_out/r06final/* (tools/r06_final_profiles.sh) into profiles/ under r06_ names, and rewrites the ONE measurements
table of DESIGN.md section 5 (between the MEASUREMENTS markers) and the headline numbers of README.md from them."""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(ROOT, "gpurun_out", "r06final"), os.path.join(ROOT, "profiles")

for w in ("cfg2", "cfg3", "cfg4"):
    shutil.copy(os.path.join(O, "pmc_summary_%s.txt" % w), os.path.join(P, "r06_%s_pmc_summary.txt" % w))
for w in ("cfg2", "cfg3", "cfg4"):
    shutil.copy(os.path.join(O, "kernel_stats_%s.csv" % w), os.path.join(P, "r06_kernel_stats_%s.csv" % w))
    shutil.copy(os.path.join(O, "kernel_trace_head_%s.csv" % w), os.path.join(P, "r06_kernel_trace_head_%s.csv" % w))
for w in ("cfg2", "cfg2_steps20", "cfg3", "cfg4", "cfg5", "hq48", "hq44", "dn8", "cfg2_s16", "n2_sharedgpu_gloo"):
    shutil.copy(os.path.join(O, "bench_%s.json" % w), os.path.join(P, "r06_bench_%s.json" % w))
for f in ("host_paths", "kseg_ab", "trace_timed_means", "lines"):
    text = [l for l in open(os.path.join(O, f + ".log")) if "amdgpu.ids" not in l]
    open(os.path.join(P, "r06_%s.log" % f), "w").writelines(text)


if os.path.isdir(os.path.join(O, "preflight")):
    dst = os.path.join(P, "r06_preflight_dry")
    os.makedirs(dst, exist_ok=True)
    for f in os.listdir(os.path.join(O, "preflight")):
        shutil.copy(os.path.join(O, "preflight", f), os.path.join(dst, f))


def line(w):
    return json.loads([x for x in open(os.path.join(P, "r06_bench_%s.json" % w)) if x.startswith("{")][0])


TIMED = {}
for l in open(os.path.join(P, "r06_trace_timed_means.log")):
    m = re.match(r"(\w+)[^:]*: (\d+) dispatches.*all: mean ([\d.]+) us; timed region \[\d+, \d+\): mean ([\d.]+) us", l)
    if m:
        TIMED[m.group(1)] = (int(m.group(2)), float(m.group(3)), float(m.group(4)))


def stats(w):
    import csv
    rows = list(csv.reader(open(os.path.join(P, "r06_kernel_stats_%s.csv" % w))))
    d = dict(zip(rows[0], rows[1]))   # (the first data row: the workload's kernel, by total duration)
    return int(d["Calls"]), float(d["AverageNs"]) / 1e3


wl = {}
for l in open(os.path.join(P, "r05_all_workloads.log")):   # (round 5's box and build: these kernels did not change in round 6)
    f = l.split()
    if len(f) > 8 and f[3] == "us":
        wl[f[0]] = (f[1], float(f[2]), float(f[4]), float(f[7]))
k = lambda x: "{:,.0f}".format(round(x, -2))
rows = ["| workload | kernel | output Msamples/s | µs per launch (`bench.py`, K timed launches behind the lead-in; rocprofv3 kernel trace of the same command) | GB/s (algorithmic) | frac of 8 TB/s | VALU issuing | HBM traffic ÷ algorithmic |",
        "|---|---|---:|---:|---:|---:|---:|---:|"]
names = {"cfg2": "cfg 2 stereo 44.1→48 kHz, 3 lobes, 10 min (default, the headline)", "cfg2_steps20": "the same as the driver runs it (`--steps 20 --warmup 3`)", "hq44": "stereo 48→44.1 kHz, 8 lobes (`hq44`)", "dn8": "stereo 44.1→8 kHz, 33 slots (`dn8`)", "cfg5": "cfg 5 same, 1 hour as one launch", "cfg4": "cfg 4 8-ch 48→44.1 kHz, 10 min",
         "cfg3": "cfg 3 stereo 8→96 kHz, 8 lobes, 10 min", "hq48": "stereo 44.1→48 kHz, 8 lobes (`hq48`)", "cfg2_s16": "cfg 2 with the opt-in int16 output (`--s16`)"}
for w in ("cfg2", "cfg2_steps20", "cfg5", "cfg4", "cfg3", "hq48", "hq44", "dn8", "cfg2_s16"):
    l = line(w); r = l["roofline"]; v = l.get("roofline_valu") or {}
    us = "%.1f" % (l["ms_per_step"] * 1e3)
    if w in ("cfg2", "cfg3", "cfg4") and w in TIMED:
        us += " (rocprofv3, the timed dispatches: %.1f; all %s: %.1f)" % (TIMED[w][2], "{:,}".format(TIMED[w][0]), TIMED[w][1])
    tr = (r["traffic"] / r["algorithmic_bytes_per_launch"]) if r.get("traffic") else None
    rows.append("| %s | `%s` | %s | %s | %s | **%.3f** | %s | %s |" % (names[w], r["kernel"], k(l["value"]), us, "{:,.0f}".format(r["achieved"]), r["frac"],
                "" if not v else "%.2f" % v["frac"], "" if tr is None else "%.3f" % tr))
for w, label in (("dn2", "stereo 48→44.1 kHz (`dn2`)"), ("mono", "mono 44.1→48 kHz (`mono`)"), ("dn1", "mono 48→44.1 kHz (`dn1`)"),
                 ("hq48m", "mono 44.1→48 kHz, 8 lobes (`hq48m`)"), ("hq44m", "mono 48→44.1 kHz, 8 lobes (`hq44m`)"), ("dn8m", "mono 44.1→8 kHz (`dn8m`)"),
                 ("dn6x", "stereo 48→8 kHz (`dn6x`)"), ("dn21", "stereo 2:1 (`dn21`)"), ("dn32", "stereo 3:2 (`dn32`)"), ("up12", "12 channels 44.1→48 kHz (`up12`)"), ("ch16", "16 channels 44.1→48 kHz (`ch16`)")):
    if w in wl:
        kern, us, ms, fr = wl[w]
        rows.append("| %s | `%s` | %s | %.1f | | %.3f | | |" % (label, kern, k(ms), us, fr))
c2 = line("cfg2")
cb = c2["cpu_baseline"]
rows.append("| reference C path on this box's host (`cpu_baseline`, kind \"%s\") | | %.0f on 1 core; %.0f on %d cores (oracle driver) | | | | | |" % (cb["kind"], cb["value"], cb["all_cores"]["value"], cb["all_cores"]["cores"]))
e2e, cbk, s1 = c2["end_to_end"], c2["callback_api"], c2["strong_curve_n1"]
n2 = line("n2_sharedgpu_gloo")
runs = [l for l in open(os.path.join(P, "r06_loop_d_summary.log")) if l.startswith("full")]
tests = "%d of %d runs green (%s each)" % (sum(" rc 0 " in l for l in runs), len(runs), runs[0].split(" rc 0 ")[1].split(" in ")[0])
notes = ("\n(One box, the final build of round 6 - source id `%s`, which stamps the PMC summaries `profiles/r06_{cfg2,cfg3,cfg4}_pmc_summary.txt`; bench lines `profiles/r06_bench_*.json`, "
         "rocprofv3 kernel stats `profiles/r06_kernel_stats_*.csv`. The rows WITHOUT a GB/s figure are round 5's box and build (`profiles/r05_all_workloads.log`): their kernels did not change in round 6 "
         "- only `cr_kseg.hpp` (cfg 3) did. `parity_full_stream` true in every line. The same lease: `pytest -m gpu` in a loop of fresh processes: %s "
         "(`profiles/r06_loop_d_summary.log`); cfg 2 `end_to_end` pageable %.2f ms / page-locked %.2f ms, the reference's callback signature %.0f Msamples/s; `strong_curve_n1` (cfg 5 in the default line) %.1f us, %.3f; "
         "N = 2 ranks sharing this one GPU over gloo (validation only): %s \"aggregate\" Msamples/s, `efficiency_vs_n1` %.2f.)"
         % (c2.get("library_source_id", open(os.path.join(P, "r06_cfg2_pmc_summary.txt")).readline().split()[-1]), tests, e2e["pageable_ms"], e2e["pinned_ms"], cbk["Msamples/s"],
            s1["ms_per_step"] * 1e3, s1["frac"], k(n2["value"]), n2["efficiency_vs_n1"]["value"]))
table = "\n".join(rows) + "\n" + notes
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
if "MEASUREMENTS_TABLE" in s:
    s = s.replace("MEASUREMENTS_TABLE", "<!-- MEASUREMENTS BEGIN (tools/r06_install_evidence.py) -->\n" + table + "\n<!-- MEASUREMENTS END -->")
else:
    s = re.sub(r"<!-- MEASUREMENTS BEGIN.*?<!-- MEASUREMENTS END -->", lambda m: "<!-- MEASUREMENTS BEGIN (tools/r06_install_evidence.py) -->\n" + table + "\n<!-- MEASUREMENTS END -->", s, flags=re.S)
open(p, "w").write(s)
print(table)
# README headline: this round's evidence box
p = os.path.join(ROOT, "README.md")
s = open(p).read()
s = re.sub(r"this round's evidence box: [^,]+,\s*[^ ]+ Msamples/s", "this round's evidence box: %.3f,\n%s Msamples/s" % (c2["roofline"]["frac"], "{:,.0f}".format(c2["value"])), s)
open(p, "w").write(s)
print("\ncfg2: %.1f us, frac %.3f, value %.0f" % (c2["ms_per_step"] * 1e3, c2["roofline"]["frac"], c2["value"]))
