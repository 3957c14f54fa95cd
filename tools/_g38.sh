set -e
O=gpurun_out/r2; mkdir -p $O
L=$O/lzf_noise.log; rm -f $L
P="timeout -k 10 200 python tools/perf_probe.py --alg none --iters 2"
for d in random mixed; do
echo "== lzf $d 64K x 65536: lanes off / default" >> $L
CW_LZF_LANES=0 $P --comp lzf --data $d --bs 65536 --nb 65536 >> $L 2>&1
$P --comp lzf --data $d --bs 65536 --nb 65536 >> $L 2>&1
echo "== lzf $d 4K x 1Mi: lanes off / default" >> $L
CW_LZF_LANES=0 $P --comp lzf --data $d --bs 4096 --nb 1048576 >> $L 2>&1
$P --comp lzf --data $d --bs 4096 --nb 1048576 >> $L 2>&1
done
echo "== lz4 mixed 4K x 1Mi: lanes off / default" >> $L
CW_LZ4_LANES=0 $P --comp lz4 --data mixed --bs 4096 --nb 1048576 >> $L 2>&1
$P --comp lz4 --data mixed --bs 4096 --nb 1048576 >> $L 2>&1
grep -v amdgpu.ids $L | sed 's/lib=libcwhc.so alg=none //; s/ | kernel ms.*//'
