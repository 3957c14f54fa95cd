set -e
O=gpurun_out/r2; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "lane or lz4 or fuzz or roundtrip" > $O/pytest_cc.log 2>&1 || { tail -30 $O/pytest_cc.log; exit 1; }
tail -2 $O/pytest_cc.log
L=$O/concurrent.log; rm -f $L
P="python tools/perf_probe.py --alg none --comp lz4"
echo "text 64K 65536 blocks: sequential lanes / concurrent reserve 24576 / 12288 / 36864" >> $L
CW_LANES_CONCURRENT=0 $P --data text --bs 65536 --nb 65536 >> $L 2>&1
$P --data text --bs 65536 --nb 65536 >> $L 2>&1
CW_LANES_RESERVE=12288 $P --data text --bs 65536 --nb 65536 >> $L 2>&1
CW_LANES_RESERVE=36864 $P --data text --bs 65536 --nb 65536 >> $L 2>&1
echo "text 64K 262144 blocks: sequential / concurrent" >> $L
CW_LANES_CONCURRENT=0 $P --data text --bs 65536 --nb 262144 --iters 2 >> $L 2>&1
$P --data text --bs 65536 --nb 262144 --iters 2 >> $L 2>&1
echo "text 64K 32768 blocks: off / concurrent" >> $L
CW_LZ4_LANES=0 $P --data text --bs 65536 --nb 32768 >> $L 2>&1
$P --data text --bs 65536 --nb 32768 >> $L 2>&1
echo "text 4K 1Mi blocks: off / small concurrent wpc 8, 16 / reserve 98304" >> $L
$P --data text --bs 4096 --nb 1048576 >> $L 2>&1
CW_LANES_SMALL=1 CW_LANES_WPC=8 $P --data text --bs 4096 --nb 1048576 >> $L 2>&1
CW_LANES_SMALL=1 CW_LANES_WPC=16 $P --data text --bs 4096 --nb 1048576 >> $L 2>&1
CW_LANES_SMALL=1 CW_LANES_WPC=16 CW_LANES_RESERVE=98304 $P --data text --bs 4096 --nb 1048576 >> $L 2>&1
echo "mixed 64K 65536" >> $L
$P --data mixed --bs 65536 --nb 65536 >> $L 2>&1
grep -v amdgpu.ids $L
