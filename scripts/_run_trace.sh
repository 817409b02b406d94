export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_latency.py -m gpu -q -x --timeout=300 -p no:cacheprovider 2>&1 | tail -n 4
DCS_LAT_EXP_STAGES=0,255,127 timeout 600 python scripts/gpu_lat_exp.py 2>&1 | grep -E "^stages|HIP"
DCS_LAT_TRACE_STAGES=255 DCS_LIB=$PWD/deepconvsep_amd/_exp_lattrace.so timeout 300 python scripts/gpu_lat_trace.py 2>&1 | grep -A2 "^ifft\|^stft"
