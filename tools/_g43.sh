set -e
O=gpurun_out/r2; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_full2.log 2>&1 || { tail -40 $O/pytest_full2.log; exit 1; }
tail -2 $O/pytest_full2.log
L=$O/final_lzf.log; rm -f $L
CW_HOST_CHUNK_MB=256 timeout -k 10 300 python tools/host_path_probe.py --data corpus --hash sha256mb --comp lzf --bs 4096 --passes 3 >> $L 2>&1
P="timeout -k 10 200 python tools/perf_probe.py --alg none --iters 2"
for d in random mixed text; do
$P --comp lzf --data $d --bs 65536 --nb 65536 >> $L 2>&1
$P --comp lzf --data $d --bs 4096 --nb 1048576 >> $L 2>&1
done
for leg in "sha256mb lzf 4096" "sha256mb lzf 65536"; do set -- $leg; python bench.py --no-legs --no-cpu-baseline --hash $1 --comp $2 --block-bytes $3 --data corpus --blocks-per-gpu $(( (4<<30) / $3 )) --steps 2 --warmup 1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('corpus $1 $2 $3', d['value'], 'GB/s ratio', d['compression_ratio'], d['roundtrip']['ok'], d['kernels']['comp']['name'])" >> $L; done
grep -v amdgpu.ids $L | sed 's/lib=libcwhc.so alg=none //; s/ | kernel ms.*//'
