#!/bin/bash
# Per-workload evidence, on the GPU box:  tools/collect_workload_profiles.sh <outdir> <workload>...
#   <outdir>/<wl>_bench.json          bench.py --workload <wl>  (default K/W)
#   <outdir>/<wl>_kernel_stats.csv    rocprofv3 --kernel-trace --stats of the same command (ALL launches: settle + warm-up included)
#   <outdir>/<wl>_kernel_timed.json   the same trace, the TIMED launches only (tools/kernel_trace_timed.py): the bench line's clock
#   <outdir>/<wl>_pmc.json            HBM bytes per step from two separate --pmc passes (FETCH_SIZE, WRITE_SIZE)
# Copy the files into profiles/rNN/ afterwards (gpurun_out/ is scratch).
set -e
OUT=$1; shift
R=$(pwd)
export TMPDIR=/tmp
mkdir -p $OUT
declare -A ALG=( [c1]=$((12*(1<<28))) [c3]=$((8*(1<<28))) [c4]=$((14*32*(1<<24))) [c5]=$(( (4*3+8)*(1<<27)/3 )) [iir]=$((8*(1<<26))) [c2]=$((16*(1<<28))) [tags]=$((1<<28)) )
PS=20; PW=5; PSETTLE=5; TOTAL=$((PS+PW+PSETTLE))
for WL in "$@"; do
  EXTRA=""
  if [ "$WL" = "c4" ]; then EXTRA="--streams 32"; fi
  python3 bench.py --workload $WL $EXTRA --no-cpu > $OUT/${WL}_bench.json 2> $OUT/${WL}_bench.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/${WL}_prof -o run -- python3 bench.py --workload $WL $EXTRA --no-cpu > /dev/null 2> $OUT/${WL}_prof.err
  find $OUT/${WL}_prof -name "*kernel_stats.csv" -exec cp {} $OUT/${WL}_kernel_stats.csv \;
  # bench.py's defaults: 40 settle + 30 warm-up steps in front of 100 timed ones (c2: + 10 launches with per-launch events behind them)
  DT=170; if [ "$WL" = "c2" ]; then DT=180; fi
  python3 tools/kernel_trace_timed.py $OUT/${WL}_kernel_timed.json $OUT/${WL}_prof 70 100 $DT > /dev/null
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/${WL}_pmc_fetch -o run -- python3 bench.py --workload $WL $EXTRA --steps $PS --warmup $PW --settle $PSETTLE --no-cpu > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/${WL}_pmc_write -o run -- python3 bench.py --workload $WL $EXTRA --steps $PS --warmup $PW --settle $PSETTLE --no-cpu > /dev/null 2>&1
  T=$TOTAL; if [ "$WL" = "c2" ]; then T=$((TOTAL+10)); fi   # c2 adds an untimed pass of min(steps, 10) launches with per-launch events
  python3 tools/pmc_workload_json.py $OUT/${WL}_pmc.json $WL ${ALG[$WL]} $T $OUT/${WL}_pmc_fetch $OUT/${WL}_pmc_write $OUT/${WL}_kernel_stats.csv > $OUT/${WL}_pmc.txt
  rm -rf $OUT/${WL}_prof $OUT/${WL}_pmc_fetch $OUT/${WL}_pmc_write
  echo "$WL done"; tail -3 $OUT/${WL}_pmc.txt
done
