set -e
O=gpurun_out/r2; mkdir -p $O
L=$O/lzf_noise2.log; rm -f $L
P="timeout -k 10 200 python tools/perf_probe.py --alg none --iters 2 --comp lzf --data mixed --bs 4096 --nb 1048576"
echo "== lzf mixed 4K x 1Mi: lanes off round 8192 / lanes off default / beside round 32768 / lanes only / beside wpc 1 / beside reserve 262144" >> $L
CW_LZF_LANES=0 CW_LZF_ROUND=8192 $P >> $L 2>&1
CW_LZF_LANES=0 $P >> $L 2>&1
CW_LZF_ROUND=32768 $P >> $L 2>&1
CW_LANES_CONCURRENT=0 $P >> $L 2>&1
CW_LANES_WPC=1 $P >> $L 2>&1
CW_LANES_RESERVE=262144 $P >> $L 2>&1
grep -v amdgpu.ids $L | sed 's/lib=libcwhc.so alg=none //; s/ | kernel ms.*//'
