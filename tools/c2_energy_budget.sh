#!/bin/bash
# Energy budget of the fused config-2 kernel: every ablation build of abl/c2_abl_<mask>/ (made by
# cariboulite_amd/_build.py build_hip_variant with -DCLHIP_RX_ABL=<mask>; masks documented in clhip_rx_pipe.hip) is run
# like the bench (same buffers, same launch) while rocm-smi samples socket power and shader clock; joules per launch =
# median power x kernel time.  Results of the ablated builds are invalid by construction: timing only.
# usage (GPU box): tools/c2_energy_budget.sh <outdir> [masks...]
OUT=${1:-gpurun_out/c2_energy}; shift
MASKS=${@:-"0 1 2 4 8 16 32 64 128 256 512 9 514 591 160 1023"}
mkdir -p $OUT
rocm-smi --showpower 2>&1 | grep -E "Power \(W\)" | sed -e 's/.*: //' > $OUT/idle_power.txt
for M in $MASKS; do
  LIB=""; if [ "$M" != "0" ]; then LIB=abl/c2_abl_$M/libcariboulite_hip.so; [ -f $LIB ] || { echo "missing $LIB"; continue; }; fi
  CLHIP_LIB=$LIB STEPS=${STEPS:-3000} timeout -k 10 300 tools/power_probe.sh c2abl_$M > $OUT/abl_$M.txt 2>&1
  cat $OUT/abl_$M.txt | tail -1
  cp gpurun_out/power_c2abl_$M.json $OUT/ 2>/dev/null
done
python3 tools/c2_energy_table.py $OUT > $OUT/table.md
cat $OUT/table.md
