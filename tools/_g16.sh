
python - <<'PY'
import numpy as np
rng = np.random.default_rng(1)
blk = [rng.integers(0, 256, 64 << 20, dtype=np.uint8).tobytes() for _ in range(4)]
with open("/tmp/random16g.bin", "wb") as f:
    for i in range(256): f.write(blk[i % 4])
PY
H=./compute_war_amd/host/hashandcompress
for c in 1 2; do echo "== driver random16g -c $c skein512+lz4 64K"; $H -g true -c $c -r 1 --block-size=65536 -H skein512 -C lz4 /tmp/random16g.bin; done
echo "== driver random16g -c 2 sha256mb+lzf 4K"; $H -g true -c 2 -r 8 -H sha256mb -C lzf /tmp/random16g.bin
echo "== driver random16g -c 2 skein+lz4 4K"; $H -g true -c 2 -r 8 -H skein -C lz4 /tmp/random16g.bin
rm -f /tmp/random16g.bin
python -m pytest tests -m gpu -x -q -k "driver" 2>&1 | tail -2
