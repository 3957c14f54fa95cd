set -e
P="timeout -k 10 200 python tools/perf_probe.py --alg none --iters 1"
for d in text mixed random; do
echo "== $d 4K"; CW_DEBUG_LZF=1 $P --comp lzf --data $d --bs 4096 --nb 1048576 2>&1 | grep -v amdgpu | sed 's/lib=libcwhc.so alg=none //; s/ | kernel ms.*//'
done
echo "== corpus 4K"; CW_DEBUG_LZF=1 python bench.py --no-legs --no-cpu-baseline --no-roundtrip --hash sha256mb --comp lzf --block-bytes 4096 --data corpus --blocks-per-gpu 1048576 --steps 1 --warmup 0 2>&1 | grep "lzf lanes" | tail -2
