#!/usr/bin/env python3
"""Per-dispatch means of the rocprofv3 --pmc passes for rx_pipe_fused_kernel, plus the derived
figures bench.py and DESIGN.md quote.  usage: pmc_to_json.py out.json <pass dir>...
FETCH_SIZE is doubled for gfx950 as /opt/skills/guides/MI355X_MICROARCH.md prescribes (HBM section);
FETCH_SIZE / WRITE_SIZE count kilobytes."""
import collections, csv, glob, json, os, sys

out, dirs = sys.argv[1], sys.argv[2:]
KERNEL = "rx_pipe_fused_kernel"
res = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg, dur = collections.defaultdict(list), {}
        for r in csv.DictReader(open(f)):
            if KERNEL not in r["Kernel_Name"]:
                continue
            # one row per (dispatch, counter [, dimension]): sum a counter's rows within a dispatch
            key = (r["Dispatch_Id"], r["Counter_Name"])
            agg[key].append(float(r["Counter_Value"]))
            dur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
        per = collections.defaultdict(list)
        for (disp, name), vals in agg.items():
            per[name].append(sum(vals))
        kus = sum(dur.values()) / max(len(dur), 1)
        for name, vals in per.items():
            res[name] = {"dispatches": len(vals), "mean_per_dispatch": sum(vals) / len(vals),
                         "pass": os.path.basename(d.rstrip("/")), "kernel_us_in_pass": kus}
n = 1 << 28
alg = 16.0 * n
der = {"note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (HBM section); separate --pmc passes; per dispatch of "
               "rx_pipe_fused_kernel, bench.py config 2 (2^28 samples)"}
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    rd = res["FETCH_SIZE"]["mean_per_dispatch"] * 1024 * 2
    wr = res["WRITE_SIZE"]["mean_per_dispatch"] * 1024
    der.update(hbm_read_bytes_per_launch_corrected_x2=rd, hbm_write_bytes_per_launch=wr,
               traffic_bytes_per_launch=rd + wr, algorithmic_bytes_per_launch=alg, traffic_over_algorithmic=(rd + wr) / alg)
if "GRBM_GUI_ACTIVE" in res:
    g = res["GRBM_GUI_ACTIVE"]
    clk = g["mean_per_dispatch"] / 8 / (g["kernel_us_in_pass"] * 1e-6)          # 8 XCDs each count the busy cycles
    der["effective_clock_ghz"] = clk / 1e9
    if "SQ_INSTS_VALU" in res and "SQ_WAVES" in res:
        tiles = n / 4096.0 * 4                                                   # wave-tiles per launch
        der["valu_insts_per_wave_tile"] = res["SQ_INSTS_VALU"]["mean_per_dispatch"] / tiles
        # a wave64 VALU instruction holds its SIMD for 4 cycles; 1024 SIMDs
        der["valu_busy_frac"] = res["SQ_INSTS_VALU"]["mean_per_dispatch"] * 4 / 1024 / (clk * g["kernel_us_in_pass"] * 1e-6)
    if "SQ_WAVE_CYCLES" in res:
        # SQ_WAVE_CYCLES counts in units of 4 cycles, summed over waves
        der["mean_waves_per_simd"] = res["SQ_WAVE_CYCLES"]["mean_per_dispatch"] * 4 / 1024 / (clk * g["kernel_us_in_pass"] * 1e-6)
if "SQ_LDS_BANK_CONFLICT" in res and "SQ_LDS_IDX_ACTIVE" in res:
    der["lds_bank_conflict_frac"] = res["SQ_LDS_BANK_CONFLICT"]["mean_per_dispatch"] / res["SQ_LDS_IDX_ACTIVE"]["mean_per_dispatch"]
res["_derived"] = der
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(der, indent=1))
