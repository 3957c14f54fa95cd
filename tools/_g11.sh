set -e
O=gpurun_out/r2; mkdir -p $O
python -m pytest tests/test_gpu_round2.py -x -q -k "lane_per_block" > $O/pytest_lanes.log 2>&1 || { tail -40 $O/pytest_lanes.log; exit 1; }
tail -2 $O/pytest_lanes.log
L=$O/lzf_lanes.log; rm -f $L
for leg in "65536 0" "65536 24576" "16384 0" "16384 24576"; do
  set -- $leg
  CW_LZF_LANES=$2 python bench.py --no-legs --no-cpu-baseline --hash sha256mb --comp lzf --block-bytes $1 --data corpus --blocks-per-gpu $(( (4<<30) / $1 )) --steps 2 --warmup 1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('lanes>=$2', d['config']['workload'][:50], d['value'], 'GB/s ratio', d['compression_ratio'], d['kernels']['comp']['name'])" >> $L
done
for w in 2 8; do CW_LANES_WPC=$w python bench.py --no-legs --no-cpu-baseline --hash sha256mb --comp lzf --block-bytes 65536 --data corpus --blocks-per-gpu 65536 --steps 2 --warmup 1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('wpc=$w', d['value'], 'GB/s')" >> $L; done
CW_LZF_LANES=0 python bench.py --no-legs --no-cpu-baseline --hash skein512 --comp lzf --block-bytes 65536 --data mixed --blocks-per-gpu 65536 --steps 2 --warmup 1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('mixed chain', d['value'], 'GB/s')" >> $L
python bench.py --no-legs --no-cpu-baseline --hash skein512 --comp lzf --block-bytes 65536 --data mixed --blocks-per-gpu 65536 --steps 2 --warmup 1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('mixed lanes', d['value'], 'GB/s')" >> $L
cat $L
CW_LIB=compute_war_amd/libcwhc_clock.so python tools/clock_probe.py > $O/clock_probe.json 2> $O/clock_probe.err || tail -5 $O/clock_probe.err
cat $O/clock_probe.json
