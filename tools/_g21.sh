set -e
O=gpurun_out/r2; mkdir -p $O
L=$O/tagged2.log; rm -f $L
P="python tools/perf_probe.py --alg none --iters 5"
for nb in 32768 65536 131072 262144 1048576; do
echo "== lzf text 4K nb=$nb: off / lanes only wpc4 / lanes only wpc8 / beside round 8192 / beside wpc2 / wpc3" >> $L
CW_LZF_LANES=0 $P --comp lzf --data text --bs 4096 --nb $nb >> $L 2>&1
CW_LZF_LANES=1 CW_LANES_CONCURRENT=0 $P --comp lzf --data text --bs 4096 --nb $nb >> $L 2>&1
CW_LZF_LANES=1 CW_LANES_CONCURRENT=0 CW_LANES_WPC=8 $P --comp lzf --data text --bs 4096 --nb $nb >> $L 2>&1
CW_LZF_LANES=1 CW_LZF_ROUND=8192 $P --comp lzf --data text --bs 4096 --nb $nb >> $L 2>&1
CW_LZF_LANES=1 CW_LANES_WPC=2 $P --comp lzf --data text --bs 4096 --nb $nb >> $L 2>&1
CW_LZF_LANES=1 CW_LANES_WPC=3 $P --comp lzf --data text --bs 4096 --nb $nb >> $L 2>&1
done
for nb in 32768 65536 131072 1048576; do
echo "== lz4 text 4K nb=$nb: off / lanes only wpc8 / default / beside reserve 16384" >> $L
CW_LZ4_LANES=0 $P --comp lz4 --data text --bs 4096 --nb $nb >> $L 2>&1
CW_LZ4_LANES=1 CW_LANES_CONCURRENT=0 $P --comp lz4 --data text --bs 4096 --nb $nb >> $L 2>&1
$P --comp lz4 --data text --bs 4096 --nb $nb >> $L 2>&1
CW_LZ4_LANES=1 CW_LANES_RESERVE=16384 $P --comp lz4 --data text --bs 4096 --nb $nb >> $L 2>&1
done
grep -v amdgpu.ids $L | sed 's/lib=libcwhc.so alg=none //; s/marked=0 | kernel ms.*//'
