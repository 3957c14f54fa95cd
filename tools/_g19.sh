set -e
O=gpurun_out/r2; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "lane or lzf or fuzz or roundtrip or lz4" > $O/pytest_cc.log 2>&1 || { tail -30 $O/pytest_cc.log; exit 1; }
tail -2 $O/pytest_cc.log
L=$O/concurrent2.log; rm -f $L
P="python tools/perf_probe.py --alg none"
echo "lzf text 4K 1Mi: off / beside wpc 4 / 8 / 16" >> $L
CW_LZF_LANES=0 $P --comp lzf --data text --bs 4096 --nb 1048576 >> $L 2>&1
$P --comp lzf --data text --bs 4096 --nb 1048576 >> $L 2>&1
CW_LANES_WPC=8 $P --comp lzf --data text --bs 4096 --nb 1048576 >> $L 2>&1
CW_LANES_WPC=16 $P --comp lzf --data text --bs 4096 --nb 1048576 >> $L 2>&1
echo "lz4 text 4K 1Mi default / 256Ki blocks" >> $L
$P --comp lz4 --data text --bs 4096 --nb 1048576 >> $L 2>&1
$P --comp lz4 --data text --bs 4096 --nb 262144 >> $L 2>&1
echo "lz4 64K mixed / text default" >> $L
$P --comp lz4 --data mixed --bs 65536 --nb 65536 >> $L 2>&1
$P --comp lz4 --data text --bs 65536 --nb 65536 >> $L 2>&1
echo "corpus legs" >> $L
for leg in "sha256mb lzf 4096" "skein lz4 4096" "sha256mb lzf 2048"; do set -- $leg; python bench.py --no-legs --no-cpu-baseline --hash $1 --comp $2 --block-bytes $3 --data corpus --blocks-per-gpu $(( (4<<30) / $3 )) --steps 2 --warmup 1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1 $2 $3', d['value'], 'GB/s ratio', d['compression_ratio'], d['kernels']['comp']['name'])" >> $L; done
grep -v amdgpu.ids $L
