#!/bin/bash
# A/B builds of the config-5 chain kernel (abl/<name>, built with _build.build_hip_variant), one line each
for V in ${VARIANTS:-"" tx_store_direct txC txH txI txJ txK txL txD txF ""}; do LIB=""; [ "$V" != "shipped" ] && [ -n "$V" ] && LIB=abl/$V/libcariboulite_hip.so; printf "%-16s " "${V:-shipped}"; env CLHIP_LIB=$LIB BENCH_TX_WARM=50 BENCH_TX_STEPS=200 python tools/bench_tx.py 2>/dev/null | tail -1; done
