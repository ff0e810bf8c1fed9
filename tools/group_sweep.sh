#!/bin/bash
# cl_group_readStream's knobs on one box (FIR64 + 3/2, 32 streams, default route): sub-batch size x (kernel stores into the mapped mirror | device buffer + copy engine) x copy threads
# usage: tools/group_sweep.sh > gpurun_out/r04/group_sweep.txt
for sink in mapped copy; do for sub in 2 4 8 16; do for th in 4 8; do
  r=$(python tools/bench_group.py --cases cf32_fir64_rs_3_2 --modes default --reps 2 --sub $sub --threads $th --sink $sink 2>/dev/null |
      python -c "import json,sys; d=json.load(sys.stdin)['cf32_fir64_rs_3_2']; s=d['default']['stats']; print(d['default']['msps_in'], d['default']['ms_per_group_call'], s['launches']//s['calls'], s['last_queue_us'], s['last_arrive_us'], s['last_total_us'], d.get('roofline',{}).get('peak'))")
  echo "sink=$sink sub=$sub threads=$th : Msps ms/call launches/call last_call(queue,arrive,total)us pcie_peak = $r"
done; done; done
