#!/usr/bin/env python3
"""Per-dispatch means of rocprofv3 --pmc passes for the kernels whose name contains <substr>, with a few derived ratios.
usage: sq_summ.py out.json <kernel substring> <pass dir>...   |   sq_summ.py --rederive in.json out.json"""
import collections, csv, glob, json, os, sys


def collect(sub, dirs):
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
    return res


def derive(res):
    """Ratios from the per-dispatch means.  Counter instances are SUMMED per dispatch: SQ_WAVE_CYCLES over the chip's 1 024 SIMDs in
    units of four cycles; GRBM_GUI_ACTIVE over its 8 XCDs; SQ_BUSY_CYCLES over its 32 shader engines.  Resident waves per SIMD =
    wave-cycles / (SIMDs x elapsed cycles).  (Round 3 divided WAVE_CYCLES / BUSY_CYCLES by 4 where the instance counts make it 8:
    its 7.8 "waves per SIMD" for a 128-VGPR kernel, which cannot hold more than 4, was twice the real 3.9.)"""
    m = lambda k: res[k]["mean_per_dispatch"] if k in res else None
    der = {}
    N_SIMD, N_XCD, N_SE = 1024, 8, 32
    if m("SQ_WAVE_CYCLES") and m("GRBM_GUI_ACTIVE"):
        der["mean_resident_waves_per_simd"] = m("SQ_WAVE_CYCLES") * 4 / N_SIMD / (m("GRBM_GUI_ACTIVE") / N_XCD)
    elif m("SQ_WAVE_CYCLES") and m("SQ_BUSY_CYCLES"):
        der["mean_resident_waves_per_simd"] = m("SQ_WAVE_CYCLES") * 4 / N_SIMD / (m("SQ_BUSY_CYCLES") / N_SE)
    if "mean_resident_waves_per_simd" in der:
        assert der["mean_resident_waves_per_simd"] <= 8.0 + 1e-6, "more waves than a gfx950 SIMD has slots: instance counts wrong?"
    if m("SQ_ACTIVE_INST_VALU") and m("SQ_WAVE_CYCLES"):
        # SQ_ACTIVE_INST_VALU: quad-cycles a SIMD's VALU is executing, summed over SIMDs
        der["valu_active_over_wave_cycles"] = m("SQ_ACTIVE_INST_VALU") / m("SQ_WAVE_CYCLES")
        if m("GRBM_GUI_ACTIVE"):
            der["valu_busy_frac_of_simd_time"] = m("SQ_ACTIVE_INST_VALU") * 4 / N_SIMD / (m("GRBM_GUI_ACTIVE") / N_XCD)
    if m("SQ_INSTS_VALU") and m("SQ_WAVES"):
        der["valu_insts_per_wave"] = m("SQ_INSTS_VALU") / m("SQ_WAVES")
    for k in ("SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_VMEM", "SQ_ACTIVE_INST_VMEM"):
        if m(k) and m("SQ_WAVE_CYCLES"):
            der[k + "_over_wave_cycles"] = m(k) / m("SQ_WAVE_CYCLES")
    return der


if __name__ == "__main__":
    if sys.argv[1] == "--rederive":          # sq_summ.py --rederive in.json out.json : the derived block again, from the stored means
        res = {k: v for k, v in json.load(open(sys.argv[2])).items() if k != "_derived"}
        res["_derived"] = derive(res)
        res["_derived"]["rederived_from"] = sys.argv[2]
        json.dump(res, open(sys.argv[3], "w"), indent=1)
        print(json.dumps(res["_derived"], indent=1))
    else:
        res = collect(sys.argv[2], sys.argv[3:])
        res["_derived"] = derive(res)
        json.dump(res, open(sys.argv[1], "w"), indent=1)
        print(json.dumps(res["_derived"], indent=1))
