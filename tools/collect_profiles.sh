#!/bin/bash
# Collect the evidence DESIGN.md section 5 cites, on the GPU box:  tools/collect_profiles.sh <tag>
#   1. bench.py (default arguments)                         -> gpurun_out/<tag>_bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command -> gpurun_out/<tag>_prof/
#   3. PMC passes, each in its own run (HBM bytes, SQ activity, LDS)   -> gpurun_out/<tag>_pmc_*/
#   4. tools/pmc_to_json.py                                 -> gpurun_out/<tag>_pmc.json
# Copy the summaries into profiles/rNN/ afterwards (gpurun_out/ is scratch).
set -e
TAG=${1:-x}
R=$(pwd)
export TMPDIR=/tmp
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o run -- python3 bench.py --no-cpu > gpurun_out/${TAG}_bench_prof.json 2> gpurun_out/${TAG}_bench_prof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -o run -- python3 bench.py --steps 20 --warmup 10 --no-cpu > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -o run -- python3 bench.py --steps 20 --warmup 10 --no-cpu > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/${TAG}_pmc_sq1 -o run -- python3 bench.py --steps 40 --warmup 30 --no-cpu > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/${TAG}_pmc_sq2 -o run -- python3 bench.py --steps 40 --warmup 30 --no-cpu > /dev/null 2>&1
python3 tools/pmc_to_json.py gpurun_out/${TAG}_pmc.json gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write gpurun_out/${TAG}_pmc_sq1 gpurun_out/${TAG}_pmc_sq2
find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
