set -e
mkdir -p gpurun_out/r2
python -m pytest tests -m gpu -x -q -k "lz4 or fuzz or roundtrip or codec or compress" > gpurun_out/r2/pytest_lz4.log 2>&1 || { tail -30 gpurun_out/r2/pytest_lz4.log; exit 1; }
tail -2 gpurun_out/r2/pytest_lz4.log
L=gpurun_out/r2/fp1.log
CW_LZ4_PARSE=v2 python tools/perf_probe.py --alg none --comp lz4 --data text --bs 65536 --nb 16384 >> $L 2>&1
for hw in 8 16 32 64; do CW_LZ4_HEADW=$hw python tools/perf_probe.py --alg none --comp lz4 --data text --bs 65536 --nb 16384 >> $L 2>&1; done
for w in 6 4 2; do CW_PARSE_WPC=$w python tools/perf_probe.py --alg none --comp lz4 --data text --bs 65536 --nb 16384 >> $L 2>&1; done
CW_LZ4_PARSE=v2 python tools/perf_probe.py --alg none --comp lz4 --data text --bs 16384 --nb 65536 >> $L 2>&1
python tools/perf_probe.py --alg none --comp lz4 --data text --bs 16384 --nb 65536 >> $L 2>&1
CW_LZ4_PARSE=v2 python tools/perf_probe.py --alg none --comp lz4 --data mixed --bs 65536 --nb 16384 >> $L 2>&1
python tools/perf_probe.py --alg none --comp lz4 --data mixed --bs 65536 --nb 16384 >> $L 2>&1
grep "lib=" $L
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for gen in v2 fp; do for c in FETCH_SIZE WRITE_SIZE; do
  CW_LZ4_PARSE=$gen timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/r2/pmc_${gen}_$c -o p -f csv -- python3 tools/perf_probe.py --alg none --comp lz4 --data text --bs 65536 --nb 16384 --iters 1 > gpurun_out/r2/pmc_${gen}_$c.log 2>&1
done; done
ls gpurun_out/r2
