#!/usr/bin/env python3
"""Summarise tools/c2_energy_budget.sh: per ablation build, kernel time, socket power, clock, joules per launch and the
difference to the full kernel (what the removed work costs)."""
import glob, os, re, sys
out = sys.argv[1]
NAMES = {0: "full kernel", 1: "no unpack (bit-cast words)", 2: "FIR window from registers (no ds_read_b128)", 4: "one tap per block (no scalar tap windows)",
         8: "no staging ds_write_b128", 16: "stores without LDS transposes", 32: "no stores, no transposes", 64: "no resampler FMAs (1 of 8 taps)",
         128: "no input loads", 256: "no workgroup barriers", 512: "no FIR FMAs (middle blocks: 3/4 of them)", 9: "no unpack + no staging writes",
         514: "no window reads + no FIR FMAs", 591: "1+2+4+8+64+512: loads -> ... -> stores only", 160: "no loads, no stores", 1023: "everything off"}
idle = 240.0
try:
    idle = float(re.findall(r"[\d.]+", open(os.path.join(out, "idle_power.txt")).read())[0])
except Exception:
    pass
rows = {}
for f in glob.glob(os.path.join(out, "abl_*.txt")):
    m = int(re.search(r"abl_(\d+)\.txt", f).group(1))
    t = open(f).read().strip().splitlines()
    if not t:
        continue
    r = re.search(r"ms/step ([\d.]+) kernel_ms ([\d.]+) samples (\d+) sclk median (\d+) MHz power median (\d+) W", t[-1])
    if r:
        rows[m] = dict(ms=float(r.group(1)), kms=float(r.group(2)), sclk=int(r.group(4)), watts=int(r.group(5)))
if 0 not in rows:
    sys.exit("no full-kernel row")
base = rows[0]
bj = base["watts"] * base["kms"] * 1e-3
print(f"idle socket power before the runs: {idle:.0f} W; full kernel {base['kms']:.4f} ms at {base['watts']} W, {base['sclk']} MHz = {bj:.3f} J per launch "
      f"({(base['watts'] - idle) * base['kms'] * 1e-3:.3f} J above idle)\n")
print("| build (CLHIP_RX_ABL) | kernel ms | W | MHz | J per launch | J saved vs full | J saved above idle |")
print("|---|---|---|---|---|---|---|")
for m in sorted(rows):
    r = rows[m]
    j = r["watts"] * r["kms"] * 1e-3
    ja = (r["watts"] - idle) * r["kms"] * 1e-3
    print(f"| {m}: {NAMES.get(m, '')} | {r['kms']:.4f} | {r['watts']} | {r['sclk']} | {j:.3f} | {bj - j:+.3f} | {(base['watts'] - idle) * base['kms'] * 1e-3 - ja:+.3f} |")
