set -e
mkdir -p gpurun_out/r2
python tests/debug_lz4_diff.py 65536 2>&1 | tail -3
python tests/debug_lz4_diff.py 8192 2>&1 | tail -2
python -m pytest tests -m gpu -x -q -k "lz4 or fuzz or roundtrip or codec or compress or driver" > gpurun_out/r2/pytest_lz4.log 2>&1 || { tail -30 gpurun_out/r2/pytest_lz4.log; exit 1; }
tail -2 gpurun_out/r2/pytest_lz4.log
L=gpurun_out/r2/fp2.log; rm -f $L
for hw in 16 32; do CW_LZ4_HEADW=$hw python tools/perf_probe.py --alg none --comp lz4 --data text --bs 65536 --nb 16384 >> $L 2>&1; done
for w in 6 4 2; do CW_PARSE_WPC=$w python tools/perf_probe.py --alg none --comp lz4 --data text --bs 65536 --nb 16384 >> $L 2>&1; done
python tools/perf_probe.py --alg none --comp lz4 --data text --bs 16384 --nb 65536 >> $L 2>&1
python tools/perf_probe.py --alg none --comp lz4 --data mixed --bs 65536 --nb 16384 >> $L 2>&1
grep "lib=" $L
