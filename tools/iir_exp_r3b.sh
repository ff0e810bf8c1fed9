set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_iir_tx.py tests/test_gpu_soapy.py -x -q -k "iir or filter" 2>&1 | tail -3
B="timeout -k 10 120 python tools/bench_iir.py"
echo "# default (eps 1e-12, b16 writes)"; CLHIP_IIR_VERBOSE=1 $B 26 30 2>&1 | tail -2; $B 26 30
echo "# eps 1e-18"; CLHIP_IIR_HORIZON_EPS=1e-18 $B 26 30
echo "# fc 10k, 25k, 100k"; $B 26 30 10e3; $B 26 30 25e3; $B 26 30 100e3
echo "# 2^17, 2^22"; $B 17 200; $B 22 100
