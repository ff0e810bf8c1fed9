set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_iir_tx.py tests/test_gpu_soapy.py -x -q -k "iir or filter" 2>&1 | tail -2
B="timeout -k 10 120 python tools/bench_iir.py"
echo "# shipped"; $B 26 30; $B 26 30
echo "# previous commit (abl/iir_prev)"; CLHIP_LIB=abl/iir_prev/libcariboulite_hip.so $B 26 30; CLHIP_LIB=abl/iir_prev/libcariboulite_hip.so $B 26 30
echo "# shipped"; $B 26 30
echo "# 2^17, 2^22"; $B 17 200; $B 22 100
python bench.py --workload iir --no-cpu 2>/dev/null | tail -1 | cut -c1-200
