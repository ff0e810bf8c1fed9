import csv, collections, sys, glob
for d in sys.argv[1:]:
    for f in glob.glob(d+"/*/*_counter_collection.csv"):
        rows=list(csv.DictReader(open(f)))
        agg=collections.defaultdict(list)
        for r in rows:
            if "rx_pipe_fused" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur=[(float(r["End_Timestamp"])-float(r["Start_Timestamp"])) for r in rows if "rx_pipe_fused" in r["Kernel_Name"]]
        print(d, "dur_us=%.1f"%(sum(dur)/len(dur)/1e3), {k:"%.4g"%(sum(v)/len(v)) for k,v in agg.items()})
