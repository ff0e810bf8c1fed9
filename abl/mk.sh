#!/bin/bash
# builds ablation variants of the shim from a patched copy of clhip_rx_pipe.hip (scratch, not committed)
set -e
R=/root/repo; C=$R/cariboulite_amd/csrc
for v in "$@"; do
  mkdir -p v$v; cp $C/clhip_rx_pipe.hip v$v/clhip_rx_pipe.hip
  python3 $R/abl/patch.py $v v$v/clhip_rx_pipe.hip
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -Wno-unused-variable -fno-slp-vectorize -I $R/include -I $C -c v$v/clhip_rx_pipe.hip -o v$v/rx.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/abl/libabl_$v.so v$v/rx.o $C/build/clhip_iir.o $C/build/clhip_runtime.o $C/build/clhip_smi.o $C/build/clhip_tx.o
done
