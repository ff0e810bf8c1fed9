import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:40]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Kind", "")) ))
ev.sort()
# the last 70 events -- or, with a kernel-name fragment as second argument, the last 70 up to the last launch of that kernel
if len(sys.argv) > 2:
    last = max(i for i, e in enumerate(ev) if sys.argv[2] in e[2])
    ev = ev[: last + 1]
tail = ev[-70:]
t0 = tail[0][0]
for s, e, n in tail:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {n}")
