#!/bin/bash
# Config 5 chain kernel: what the time is made of.  Ablation builds (abl/tx_abl_<mask>, results invalid, timing only):
# 1 = the sum pass reads sub-block 0 only, 2 = the sub-block loop reads no messages, 4 = no output stores, 8 = no look-back.
# Build here first:  python -c "from cariboulite_amd import _build; [_build.build_hip_variant(f'tx_abl_{m}', [f'TXQ_ABL={m}'], source='clhip_tx.hip') for m in (1,2,3,4,7,8,9)]"
# Run on the GPU box:  bash tools/tx_ablations.sh > gpurun_out/tx_ablations.txt
VARIANTS="shipped tx_abl_1 tx_abl_2 tx_abl_3 tx_abl_4 tx_abl_7 tx_abl_8 tx_abl_9 shipped" bash tools/tx_variants.sh
