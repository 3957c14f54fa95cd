set -e
O=gpurun_out/r2; mkdir -p $O
L=$O/tagged3.log; rm -f $L
P="python tools/perf_probe.py --alg none --iters 5"
for nb in 131072 262144 1048576; do
for rr in "8192 16384" "8192 24576" "8192 32768" "4096 8192" "4096 16384" "4096 24576" "16384 32768"; do set -- $rr
echo "== lzf text 4K nb=$nb round $1 reserve $2" >> $L
CW_LZF_LANES=1 CW_LZF_ROUND=$1 CW_LANES_RESERVE=$2 $P --comp lzf --data text --bs 4096 --nb $nb >> $L 2>&1
done
for rs in 4096 8192 16384 32768; do
echo "== lz4 text 4K nb=$nb reserve $rs" >> $L
CW_LZ4_LANES=1 CW_LANES_RESERVE=$rs $P --comp lz4 --data text --bs 4096 --nb $nb >> $L 2>&1
done
done
grep -v amdgpu.ids $L | sed 's/lib=libcwhc.so alg=none //; s/marked=0 | kernel ms.*//'
