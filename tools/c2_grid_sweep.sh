#!/bin/bash
# Fused RX kernel, config 2: how many workgroups walk the tiles.  Default = persistent workers (what is resident, 4 per CU)
# on the tile queue; CLHIP_QUEUE_K=0 = static striding with CLHIP_WG_PER_CU workgroups per CU, up to one tile per workgroup
# (259 x 256 > the 66 052 tiles of 2^28 samples: no loop, no prefetch, the dispatcher does the scheduling).
# Run on the GPU box:  bash tools/c2_grid_sweep.sh > gpurun_out/c2_grid_sweep.txt
run() { printf "%-44s " "$1"; env $1 python bench.py --no-cpu --steps 60 --warmup 30 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])"; }
run "CLHIP_QUEUE_K=2"
run "CLHIP_QUEUE_K=0"
run "CLHIP_QUEUE_K=0 CLHIP_WG_PER_CU=4"
run "CLHIP_QUEUE_K=0 CLHIP_WG_PER_CU=8"
run "CLHIP_QUEUE_K=0 CLHIP_WG_PER_CU=32"
run "CLHIP_QUEUE_K=0 CLHIP_WG_PER_CU=64"
run "CLHIP_QUEUE_K=0 CLHIP_WG_PER_CU=129"
run "CLHIP_QUEUE_K=0 CLHIP_WG_PER_CU=259"
run "CLHIP_QUEUE_K=1"
run "CLHIP_QUEUE_K=2"
