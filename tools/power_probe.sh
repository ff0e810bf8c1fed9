#!/bin/bash
# Sample socket power / shader clock with rocm-smi while bench.py's workload runs (diagnostic only).
# usage: [ENV=...] tools/power_probe.sh <tag> ; writes gpurun_out/power_<tag>.{json,txt}
TAG=${1:-run}
python bench.py --no-cpu --workload ${WORKLOAD:-c2} --steps ${STEPS:-8000} --warmup 30 > gpurun_out/power_$TAG.json 2> gpurun_out/power_$TAG.err &
BP=$!
: > gpurun_out/power_$TAG.txt
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Power \(W\)|sclk" | sed -e 's/.*: //' | tr '\n' ' ' >> gpurun_out/power_$TAG.txt
  echo >> gpurun_out/power_$TAG.txt
  sleep 0.3
done
wait $BP
python - "$TAG" <<'PY'
import sys, re, json, statistics as st
tag = sys.argv[1]
rows = [re.findall(r"[\d.]+", l) for l in open(f"gpurun_out/power_{tag}.txt")]
rows = [(float(r[0]), float(r[1])) for r in rows if len(r) >= 2]
busy = [r for r in rows if r[1] > 600]
b = json.loads(open(f"gpurun_out/power_{tag}.json").read())
print(tag, "ms/step %.4f" % b["ms_per_step"], "kernel_ms %.4f" % b["roofline"].get("kernel_ms_avg", 0), "samples", len(busy),
      "sclk median %.0f MHz" % (st.median(r[0] for r in busy) if busy else 0), "power median %.0f W" % (st.median(r[1] for r in busy) if busy else 0))
PY
