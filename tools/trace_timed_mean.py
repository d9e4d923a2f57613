#!/usr/bin/env python3
"""Mean duration of the TIMED dispatches of a bench.py run from a rocprofv3 --kernel-trace of the same command.
usage: trace_timed_mean.py <kernel_trace.csv> <bench line .json> [kernel-name substring]
bench.py says which of its launches the timed region was (`timed_dispatches`: first, count - in launch order); the rocprofv3 --stats
average runs over EVERY dispatch of the process, clock ramp included, and cannot reproduce ms_per_step."""
import csv, json, sys

trace, line = sys.argv[1], sys.argv[2]
want = sys.argv[3] if len(sys.argv) > 3 else "k_"
j = json.loads(next(l for l in open(line) if l.startswith("{")))
first, count = j["timed_dispatches"]["first"], j["timed_dispatches"]["count"]
rows = [r for r in csv.DictReader(open(trace)) if want in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
timed = dur[first:first + count]
tail = dur[first + count:]
name = rows[first]["Kernel_Name"][:60] if len(rows) > first else "?"
print("%s: %d dispatches of %s...; all: mean %.2f us; timed region [%d, %d): mean %.2f us (bench.py ms_per_step %.2f us, launch_us.median %.2f us); the %d after it: mean %.2f us"
      % (j["config"]["workload"].split(":")[0], len(dur), name, sum(dur) / max(len(dur), 1), first, first + count, sum(timed) / max(len(timed), 1), j["ms_per_step"] * 1e3,
         j["launch_us"]["median"], len(tail), sum(tail) / max(len(tail), 1)))
