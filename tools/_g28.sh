set -e
O=gpurun_out/r2; mkdir -p $O
L=$O/ring2.log; rm -f $L
P="timeout -k 10 120 python tools/perf_probe.py --alg none --iters 3"
echo "== lz4 text 64K 131072 blocks wpc 6: normal / all blocks' output to one slot" >> $L
CW_LANES_WPC=6 $P --comp lz4 --data text --bs 65536 --nb 131072 >> $L 2>&1
CW_DEBUG_NOOUT=1 CW_LANES_WPC=6 $P --comp lz4 --data text --bs 65536 --nb 131072 >> $L 2>&1
grep -v amdgpu.ids $L | sed 's/lib=libcwhc.so alg=none //; s/marked=0 | kernel ms.*//'
