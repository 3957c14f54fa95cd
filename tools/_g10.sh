O=gpurun_out/r2; mkdir -p $O
CW_LZ4_LANES=1 timeout -k 10 120 python tests/debug_lz4_diff.py 65536 2>&1 | tail -4
CW_LZ4_LANES=1 timeout -k 10 120 python tests/debug_lz4_diff.py 8192 lcet10.txt zeros:65536 kennedy.xls ptt5 sum 2>&1 | tail -4
L=$O/lanes1.log; rm -f $L
CW_LZ4_LANES=0 timeout -k 10 200 python tools/perf_probe.py --alg none --comp lz4 --data text --bs 65536 --nb 65536 >> $L 2>&1
for w in 2 4 8 16; do CW_LANES_WPC=$w timeout -k 10 200 python tools/perf_probe.py --alg none --comp lz4 --data text --bs 65536 --nb 65536 >> $L 2>&1; done
timeout -k 10 200 python tools/perf_probe.py --alg none --comp lz4 --data text --bs 16384 --nb 262144 >> $L 2>&1
timeout -k 10 200 python tools/perf_probe.py --alg none --comp lz4 --data mixed --bs 65536 --nb 65536 >> $L 2>&1
grep "lib=" $L
