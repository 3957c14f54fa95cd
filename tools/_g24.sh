set -e
O=gpurun_out/r2; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "lane or fuzz or roundtrip or lz4" > $O/pytest_fp.log 2>&1 || { tail -40 $O/pytest_fp.log; exit 1; }
tail -2 $O/pytest_fp.log
L=$O/lanefp.log; rm -f $L
P="timeout -k 10 120 python tools/perf_probe.py --alg none --iters 3"
for d in text mixed; do
echo "== lz4 $d 64K 131072 blocks: plain / fp / fp wpc 4 / wpc 12 / wpc 16" >> $L
CW_LZ4_LANES_FP=0 $P --comp lz4 --data $d --bs 65536 --nb 131072 >> $L 2>&1
$P --comp lz4 --data $d --bs 65536 --nb 131072 >> $L 2>&1
CW_LANES_WPC=4 $P --comp lz4 --data $d --bs 65536 --nb 131072 >> $L 2>&1
CW_LANES_WPC=12 $P --comp lz4 --data $d --bs 65536 --nb 131072 >> $L 2>&1
CW_LANES_WPC=16 $P --comp lz4 --data $d --bs 65536 --nb 131072 >> $L 2>&1
done
echo "== lz4 text 16K 262144 blocks plain / fp" >> $L
CW_LZ4_LANES_FP=0 $P --comp lz4 --data text --bs 16384 --nb 262144 >> $L 2>&1
$P --comp lz4 --data text --bs 16384 --nb 262144 >> $L 2>&1
grep -v amdgpu.ids $L | sed 's/lib=libcwhc.so alg=none //; s/marked=0 | kernel ms.*//'
