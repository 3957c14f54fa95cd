set -e
O=gpurun_out/r2; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "lzf or fuzz or roundtrip or pipeline or config3 or driver" > $O/pytest_lzf.log 2>&1 || { tail -40 $O/pytest_lzf.log; exit 1; }
tail -2 $O/pytest_lzf.log
for leg in "sha256mb lzf 65536 corpus" "sha256mb lzf 4096 corpus" "sha256mb lzf 16384 corpus" "skein512 lzf 65536 mixed"; do
  set -- $leg
  python bench.py --no-legs --no-cpu-baseline --hash $1 --comp $2 --block-bytes $3 --data $4 --blocks-per-gpu $(( (4<<30) / $3 )) --steps 3 --warmup 1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['config']['workload'][:60], d['value'], 'GB/s ratio', d['compression_ratio'], d['kernels']['comp']['ms_per_step'])"
done
python tools/perf_probe.py --alg none --comp lzf --data text --bs 65536 --nb 16384
python tools/perf_probe.py --alg none --comp lzf --data text --bs 4096 --nb 262144
python - <<'PY'
import numpy as np
rng = np.random.default_rng(1)
blk = [rng.integers(0, 256, 64 << 20, dtype=np.uint8).tobytes() for _ in range(4)]
with open("/tmp/random16g.bin", "wb") as f:
    for i in range(256): f.write(blk[i % 4])
PY
H=./compute_war_amd/host/hashandcompress
for c in 1 2 3 4; do echo "== random16g -c $c skein512+lz4 64K"; $H -g true -c $c -r 1 --block-size=65536 -H skein512 -C lz4 /tmp/random16g.bin; done
rm -f /tmp/random16g.bin
