set -e
mkdir -p gpurun_out/r2
python -m pytest tests -m gpu -x -q > gpurun_out/r2/pytest0.log 2>&1 || { tail -20 gpurun_out/r2/pytest0.log; exit 1; }
tail -2 gpurun_out/r2/pytest0.log
./tools/ubench > gpurun_out/r2/ubench.txt 2>&1
for w in 10 8 6 4 2; do CW_PARSE_WPC=$w python tools/perf_probe.py --alg none --comp lz4 --data text --bs 65536 --nb 16384 >> gpurun_out/r2/wpc64k.log 2>&1; done
for w in 8 4 2; do CW_PARSE_WPC=$w python tools/perf_probe.py --alg none --comp lz4 --data text --bs 4096 --nb 262144 >> gpurun_out/r2/wpc4k.log 2>&1; done
python tools/perf_probe.py --alg none --comp lzf --data text --bs 65536 --nb 16384 >> gpurun_out/r2/lzf.log 2>&1
python tools/perf_probe.py --alg none --comp lzf --data text --bs 4096 --nb 262144 >> gpurun_out/r2/lzf.log 2>&1
cat gpurun_out/r2/wpc64k.log gpurun_out/r2/wpc4k.log gpurun_out/r2/lzf.log
