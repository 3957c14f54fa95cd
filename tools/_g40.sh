set -e
O=gpurun_out/r2; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "lzf or side_by_side or lane or fuzz or roundtrip" > $O/pytest_hb.log 2>&1 || { tail -40 $O/pytest_hb.log; exit 1; }
tail -2 $O/pytest_hb.log
L=$O/lzf_noise3.log; rm -f $L
P="timeout -k 10 200 python tools/perf_probe.py --alg none --iters 2"
for d in random mixed text; do
echo "== lzf $d 64K x 65536: lanes off / default" >> $L
CW_LZF_LANES=0 $P --comp lzf --data $d --bs 65536 --nb 65536 >> $L 2>&1
$P --comp lzf --data $d --bs 65536 --nb 65536 >> $L 2>&1
echo "== lzf $d 4K x 1Mi: lanes off / default" >> $L
CW_LZF_LANES=0 $P --comp lzf --data $d --bs 4096 --nb 1048576 >> $L 2>&1
$P --comp lzf --data $d --bs 4096 --nb 1048576 >> $L 2>&1
done
for leg in "sha256mb lzf 4096" "sha256mb lzf 65536"; do set -- $leg; python bench.py --no-legs --no-cpu-baseline --hash $1 --comp $2 --block-bytes $3 --data corpus --blocks-per-gpu $(( (4<<30) / $3 )) --steps 2 --warmup 1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('corpus $1 $2 $3', d['value'], 'GB/s ratio', d['compression_ratio'], d['kernels']['comp']['name'])" >> $L; done
grep -v amdgpu.ids $L | sed 's/lib=libcwhc.so alg=none //; s/ | kernel ms.*//'
