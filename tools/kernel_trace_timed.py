#!/usr/bin/env python3
"""The bench line's clock, reproduced from a rocprofv3 --kernel-trace run of the same command: per kernel, the average duration of
the TIMED launches only -- bench.py queues `settle` + `warmup` untimed steps first (DVFS transient: the first steps of a run are up to
40 % slower), and `--stats` averages them in, which is why round 3's kernel_stats.csv sat 2-3 % above the bench line -- next to the
all-launch average, and the distance from the first timed launch's start to the last one's end per step (what HIP events around the
timed region measure: dispatch gaps included).

usage: kernel_trace_timed.py out.json <trace dir or kernel_trace.csv> <untimed steps in front> <timed steps> [<total steps of the run>]"""
import collections, csv, glob, json, os, sys

out, src, skip, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
total = int(sys.argv[5]) if len(sys.argv) > 5 else skip + steps
SKIP = ("at::native", "elementwise_kernel", "distribution", "vectorized", "Memset", "fill", "__amd_rocclr")   # torch's input generators
files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
per = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        if any(s in r["Kernel_Name"] for s in SKIP):
            continue
        per[r["Kernel_Name"].split("(")[0][:120]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
res = {"untimed_steps_in_front": skip, "timed_steps": steps, "kernels": {}}
for name, d in per.items():
    d.sort()
    lps = max(len(d) // total, 1)                         # launches of this kernel per bench step
    t = d[skip * lps:(skip + steps) * lps]
    if not t:
        continue
    res["kernels"][name] = {
        "launches": len(d), "launches_per_step": lps,
        "kernel_us_all_avg": round(sum(e - s for s, e in d) / len(d) / 1e3, 3),
        "kernel_us_all_max": round(max(e - s for s, e in d) / 1e3, 3),
        "kernel_us_timed_avg": round(sum(e - s for s, e in t) / len(t) / 1e3, 3),
        "kernel_us_timed_min": round(min(e - s for s, e in t) / 1e3, 3), "kernel_us_timed_max": round(max(e - s for s, e in t) / 1e3, 3),
        "timed_span_us_per_step": round((t[-1][1] - t[0][0]) / steps / 1e3, 3)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
