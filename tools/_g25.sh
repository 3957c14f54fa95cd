set -e
O=gpurun_out/r2; mkdir -p $O
L=$O/lanefp2.log; rm -f $L
P="timeout -k 10 120 python tools/perf_probe.py --alg none --iters 3"
echo "== lz4 mixed 64K 131072 blocks: wpc 8 / 4; 65536 blocks wpc 8; text 64K 131072" >> $L
$P --comp lz4 --data mixed --bs 65536 --nb 131072 >> $L 2>&1
CW_LANES_WPC=4 $P --comp lz4 --data mixed --bs 65536 --nb 131072 >> $L 2>&1
$P --comp lz4 --data mixed --bs 65536 --nb 65536 >> $L 2>&1
$P --comp lz4 --data text --bs 65536 --nb 131072 >> $L 2>&1
echo "== lz4 text 4K 1Mi; corpus legs" >> $L
$P --comp lz4 --data text --bs 4096 --nb 1048576 >> $L 2>&1
for leg in "skein512 lz4 65536" "sha256mb lzf 4096" "skein lz4 4096" "sha256mb lzf 65536"; do set -- $leg; python bench.py --no-legs --no-cpu-baseline --hash $1 --comp $2 --block-bytes $3 --data corpus --blocks-per-gpu $(( (4<<30) / $3 )) --steps 2 --warmup 1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1 $2 $3', d['value'], 'GB/s ratio', d['compression_ratio'], d['kernels']['comp']['name'])" >> $L; done
grep -v amdgpu.ids $L | sed 's/lib=libcwhc.so alg=none //; s/marked=0 | kernel ms.*//'
