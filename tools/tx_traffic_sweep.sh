#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-"" abl/tx_n4 abl/tx_n5 abl/tx_n8}; do
  L=""; [ -n "$v" ] && L=$GRAFT_REPO_ROOT/$v/libcariboulite_hip.so
  tag=$(basename "${v:-default}")
  export CLHIP_LIB=$L
  bash tools/collect_workload_profiles.sh gpurun_out/tx_traffic_$tag c5 > gpurun_out/tx_traffic_$tag.log 2>&1
  echo "== $tag"; grep -E "traffic_over_algorithmic|hbm_read|hbm_write" gpurun_out/tx_traffic_$tag/c5_pmc.txt; grep -E "tx_fm" gpurun_out/tx_traffic_$tag/c5_kernel_stats.csv | cut -c1-120
done
