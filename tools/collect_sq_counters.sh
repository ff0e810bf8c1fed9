#!/bin/bash
# SQ activity counters of a bench workload (what its waves do with their cycles): tools/collect_sq_counters.sh <outdir> <workload>...
set -e
OUT=$1; shift
R=$(pwd); export TMPDIR=/tmp; mkdir -p $OUT
for WL in "$@"; do
  EXTRA=""; [ "$WL" = "c4" ] && EXTRA="--streams 32"
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $R/$OUT/${WL}_sq1 -o run -- python3 bench.py --workload $WL $EXTRA --steps 20 --warmup 10 --no-cpu > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/$OUT/${WL}_sq2 -o run -- python3 bench.py --workload $WL $EXTRA --steps 20 --warmup 10 --no-cpu > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/$OUT/${WL}_sq3 -o run -- python3 bench.py --workload $WL $EXTRA --steps 20 --warmup 10 --no-cpu > /dev/null 2>&1 || true
  declare -A KN=( [c1]=smi_unpack_kernel [c2]=rx_pipe_fused_kernel [c3]=rx_pipe_fused_kernel [c4]=rx_pipe_fused_kernel [c5]=tx_fm_chain_kernel [iir]=iir_rail_kernel )
  python3 tools/sq_summ.py $OUT/${WL}_sq.json ${KN[$WL]} $OUT/${WL}_sq1 $OUT/${WL}_sq2 $OUT/${WL}_sq3
  rm -rf $OUT/${WL}_sq1 $OUT/${WL}_sq2 $OUT/${WL}_sq3
  echo "$WL done"
done
