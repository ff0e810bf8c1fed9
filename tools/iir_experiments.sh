#!/bin/bash
set -o pipefail
O=gpurun_out/r3_iir; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_iir_tx.py -x -q -k "iir" > $O/pytest_iir.log 2>&1 || { tail -30 $O/pytest_iir.log; exit 1; }
tail -2 $O/pytest_iir.log
B="timeout -k 10 120 python tools/bench_iir.py"
{
echo "# default (prio)"; $B 26 20
echo "# no prio"; CLHIP_IIR_PRIO=0 $B 26 20
echo "# prio, chunk 4 / 16"; for c in 4 16; do CLHIP_IIR_CHUNK=$c $B 26 20; done
echo "# prio, waves per CU 12 / 8"; CLHIP_IIR_WG_PER_CU=12 $B 26 20; CLHIP_IIR_WG_PER_CU=8 $B 26 20
echo "# round-2 library"; CLHIP_LIB=abl/r2/libcariboulite_hip.so $B 26 20
echo "# 2^22 / 2^24"; $B 22 50; $B 24 50
echo "# fc 10 kHz / 25 kHz at 2^26"; $B 26 20 10e3; $B 26 20 25e3
echo "# stamps"; timeout -k 10 120 python tools/iir_phase_stamps.py 26
} 2>&1 | grep -v amdgpu.ids | tee $O/bench7.log
