set -e
O=gpurun_out/r2; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "lane or fuzz or roundtrip or lz4" > $O/pytest_ring.log 2>&1 || { tail -40 $O/pytest_ring.log; exit 1; }
tail -2 $O/pytest_ring.log
L=$O/ring.log; rm -f $L
P="timeout -k 10 120 python tools/perf_probe.py --alg none --iters 3"
for d in text mixed; do
echo "== lz4 $d 64K 131072 blocks: ring0 / ring1 / ring2 / ring1 wpc 4 / wpc 6" >> $L
CW_LZ4_LANES_RING=0 $P --comp lz4 --data $d --bs 65536 --nb 131072 >> $L 2>&1
$P --comp lz4 --data $d --bs 65536 --nb 131072 >> $L 2>&1
CW_LZ4_LANES_RING=2 $P --comp lz4 --data $d --bs 65536 --nb 131072 >> $L 2>&1
CW_LANES_WPC=4 $P --comp lz4 --data $d --bs 65536 --nb 131072 >> $L 2>&1
CW_LANES_WPC=6 $P --comp lz4 --data $d --bs 65536 --nb 131072 >> $L 2>&1
done
echo "== lz4 text 16K 262144 blocks ring0 / ring1" >> $L
CW_LZ4_LANES_RING=0 $P --comp lz4 --data text --bs 16384 --nb 262144 >> $L 2>&1
$P --comp lz4 --data text --bs 16384 --nb 262144 >> $L 2>&1
grep -v amdgpu.ids $L | sed 's/lib=libcwhc.so alg=none //; s/marked=0 | kernel ms.*//'
