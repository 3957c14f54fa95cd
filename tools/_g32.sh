set -e
O=gpurun_out/r2; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "decod or round_trip or lane or driver or perf" > $O/pytest_dec2.log 2>&1 || { tail -40 $O/pytest_dec2.log; exit 1; }
tail -2 $O/pytest_dec2.log
timeout -k 10 300 python tools/decode_probe.py 4096 65536 4096 > $O/decode3.log 2>&1; grep -v amdgpu $O/decode3.log
