#!/bin/bash
set -o pipefail
O=gpurun_out/r3_iir; mkdir -p $O
{
echo "# stamps dynamic"; timeout -k 10 120 python tools/phase_stamps.py 26
echo "# stamps static"; CLHIP_IIR_DYNAMIC=0 timeout -k 10 120 python tools/phase_stamps.py 26
echo "# stamps static 8 per CU"; CLHIP_IIR_DYNAMIC=0 CLHIP_IIR_WG_PER_CU=8 timeout -k 10 120 python tools/phase_stamps.py 26
} 2>&1 | grep -v amdgpu.ids | tee $O/bench5.log
