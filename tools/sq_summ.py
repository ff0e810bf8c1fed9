#!/usr/bin/env python3
"""Per-dispatch means of rocprofv3 --pmc passes for the kernels whose name contains <substr>, with a few derived ratios.
usage: sq_summ.py out.json <kernel substring> <pass dir>..."""
import collections, csv, glob, json, os, sys
out, sub, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
res = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg, dur = collections.defaultdict(float), {}
        for r in csv.DictReader(open(f)):
            if sub not in r["Kernel_Name"]:
                continue
            agg[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            dur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
        per = collections.defaultdict(list)
        for (disp, name), v in agg.items():
            per[name].append(v)
        kus = sum(dur.values()) / max(len(dur), 1)
        for name, vals in per.items():
            res[name] = {"dispatches": len(vals), "mean_per_dispatch": sum(vals) / len(vals), "kernel_us_in_pass": round(kus, 2)}
m = lambda k: res[k]["mean_per_dispatch"] if k in res else None
der = {}
if m("SQ_WAVE_CYCLES") and m("SQ_BUSY_CYCLES"):
    der["mean_resident_waves_per_simd"] = m("SQ_WAVE_CYCLES") / m("SQ_BUSY_CYCLES") / 4 * 1.0
if m("SQ_ACTIVE_INST_VALU") and m("SQ_BUSY_CYCLES"):
    # SQ_ACTIVE_INST_VALU: cycles (x4: counted per quad-cycle) a SIMD's VALU is executing, summed over SIMDs; SQ_BUSY_CYCLES per SE/XCD instance
    der["valu_active_over_wave_cycles"] = m("SQ_ACTIVE_INST_VALU") / m("SQ_WAVE_CYCLES") if m("SQ_WAVE_CYCLES") else None
if m("SQ_INSTS_VALU") and m("SQ_WAVES"):
    der["valu_insts_per_wave"] = m("SQ_INSTS_VALU") / m("SQ_WAVES")
for k in ("SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_VMEM", "SQ_ACTIVE_INST_VMEM"):
    if m(k) and m("SQ_WAVE_CYCLES"):
        der[k + "_over_wave_cycles"] = m(k) / m("SQ_WAVE_CYCLES")
res["_derived"] = der
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(der, indent=1))
