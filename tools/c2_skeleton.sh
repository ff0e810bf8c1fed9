#!/bin/bash
# the fused kernel and its memory skeleton (compute stripped by ablation masks, abl/<name>): timing only
for V in "" c2_unaligned c2_abl_591 c2_abl_591u c2_abl_607 c2_abl_160 "" c2_unaligned; do
  LIB=""; [ -n "$V" ] && LIB=abl/$V/libcariboulite_hip.so
  echo -n "${V:-shipped}: "; env CLHIP_LIB=$LIB python bench.py --no-cpu --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline'].get('kernel_ms_min'))"
done
