#!/usr/bin/env python3
"""HBM traffic of one bench.py workload from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE count kilobytes and FETCH_SIZE is doubled
on gfx950).  Every kernel of the workload is summed per bench step.

usage: pmc_workload_json.py out.json <workload> <algorithmic bytes per step> <steps in the profiled runs> <fetch dir> <write dir> [<stats csv>]
"""
import collections, csv, glob, json, os, sys

out, wl, alg, steps = sys.argv[1], sys.argv[2], float(sys.argv[3]), int(sys.argv[4])
dirs = {"FETCH_SIZE": sys.argv[5], "WRITE_SIZE": sys.argv[6]}
stats = sys.argv[7] if len(sys.argv) > 7 else None
SKIP = ("at::native", "elementwise_kernel", "distribution", "vectorized", "Memset", "fill", "__amd_rocclr")   # torch's input generators
res = {"workload": wl, "steps_profiled": steps, "kernels": {}}
tot = {}
for ctr, d in dirs.items():
    per_kernel = collections.defaultdict(lambda: [0.0, set(), 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr or any(s in r["Kernel_Name"] for s in SKIP):
                continue
            k = per_kernel[r["Kernel_Name"]]
            k[0] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in k[1]:
                k[1].add(r["Dispatch_Id"])
                k[2] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    tot[ctr] = 0.0
    for name, (v, disp, us) in per_kernel.items():
        short = name.split("(")[0][:120]
        kb_per_step = v / steps
        e = res["kernels"].setdefault(short, {})
        e[ctr + "_kb_per_step"] = kb_per_step
        e["launches_per_step"] = len(disp) / steps
        e["avg_us_in_" + ctr + "_pass"] = us / max(len(disp), 1)
        tot[ctr] += kb_per_step
rd = tot["FETCH_SIZE"] * 1024 * 2           # gfx950 correction
wr = tot["WRITE_SIZE"] * 1024
res["_derived"] = {"hbm_read_bytes_per_step_corrected_x2": rd, "hbm_write_bytes_per_step": wr,
                   "traffic_bytes_per_step": rd + wr, "algorithmic_bytes_per_step": alg,
                   "traffic_over_algorithmic": (rd + wr) / alg,
                   "note": "separate --pmc passes; FETCH_SIZE x2 (gfx950) and KB units per MI355X_MICROARCH.md; all kernels of the "
                           "workload summed per bench.py step (torch's input generators excluded)"}
if stats and os.path.exists(stats):
    ks = {}
    for r in csv.DictReader(open(stats)):
        if any(s in r["Name"] for s in SKIP):
            continue
        ks[r["Name"].split("(")[0][:120]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                                "min_us": float(r["MinNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6}
    res["kernel_stats"] = ks
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["_derived"], indent=1))
