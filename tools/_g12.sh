set -e
O=gpurun_out/r2; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_full.log 2>&1 || { tail -40 $O/pytest_full.log; exit 1; }
tail -2 $O/pytest_full.log
L=$O/small_lanes.log; rm -f $L
for e in 0 1; do
 for w in 4 16; do
  CW_LANES_SMALL=$e CW_LANES_WPC=$w python bench.py --no-legs --no-cpu-baseline --hash sha256mb --comp lzf --block-bytes 4096 --data corpus --blocks-per-gpu 1048576 --steps 2 --warmup 1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('lzf 4K small=$e wpc=$w', d['value'], 'GB/s', d['kernels']['comp']['name'])" >> $L
  CW_LANES_SMALL=$e CW_LANES_WPC=$w python bench.py --no-legs --no-cpu-baseline --hash skein --comp lz4 --block-bytes 4096 --data corpus --blocks-per-gpu 1048576 --steps 2 --warmup 1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('lz4 4K small=$e wpc=$w', d['value'], 'GB/s', d['kernels']['comp']['name'])" >> $L
 done
done
for w in 8 16; do CW_LANES_WPC=$w python bench.py --no-legs --no-cpu-baseline --data corpus --blocks-per-gpu 262144 --steps 2 --warmup 1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('lz4 64K 16GiB corpus wpc=$w', d['value'], 'GB/s')" >> $L; done
cat $L
