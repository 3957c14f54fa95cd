set -e
mkdir -p gpurun_out/r2

python -m pytest tests -m gpu -x -q > gpurun_out/r2/pytest_full.log 2>&1 || { tail -40 gpurun_out/r2/pytest_full.log; exit 1; }
tail -3 gpurun_out/r2/pytest_full.log
python - <<'PY'
import os
root = "tests/golden/corpus/canterbury/"
t = b"".join(open(root + f, "rb").read() for f in ("lcet10.txt", "kennedy.xls", "ptt5"))
with open("/tmp/corpus4g.bin", "wb") as f:
    n = 0
    while n < 4 << 30:
        f.write(t); n += len(t)
import numpy as np
rng = np.random.default_rng(1)
with open("/tmp/random4g.bin", "wb") as f:
    for _ in range(64):
        f.write(rng.integers(0, 256, 64 << 20, dtype=np.uint8).tobytes())
PY
L=gpurun_out/r2/hostpath.log; rm -f $L
H=./compute_war_amd/host/hashandcompress
for f in /tmp/corpus4g.bin /tmp/random4g.bin; do
  for c in 2 4 8; do
    echo "== $f -c $c skein512+lz4 64K" >> $L; $H -v -g true -c $c -r 1 --block-size=65536 -H skein512 -C lz4 $f >> $L 2>&1
  done
  echo "== $f -c 4 skein+lz4 4K" >> $L; $H -v -g true -c 4 -r 8 -H skein -C lz4 $f >> $L 2>&1
  echo "== $f -c 4 sha256mb+lzf 4K" >> $L; $H -v -g true -c 4 -r 8 -H sha256mb -C lzf $f >> $L 2>&1
done
cat $L
./compute_war_amd/host/mgpu_stream --devices 1 --blocks-per-gpu 262144 --steps 3 --warmup 1 | tee gpurun_out/r2/mgpu1.log
