set -e
O=gpurun_out/r2; mkdir -p $O
python tools/pcie_probe.py | tee $O/pcie_probe.json
python - <<'PY'
import numpy as np
rng = np.random.default_rng(1)
blk = [rng.integers(0, 256, 64 << 20, dtype=np.uint8).tobytes() for _ in range(4)]
with open("/tmp/random16g.bin", "wb") as f:
    for i in range(256): f.write(blk[i % 4])
root = "tests/golden/corpus/canterbury/"
t = b"".join(open(root + f, "rb").read() for f in ("lcet10.txt", "kennedy.xls", "ptt5"))
with open("/tmp/corpus4g.bin", "wb") as f:
    n = 0
    while n < 4 << 30:
        f.write(t); n += len(t)
PY
H=./compute_war_amd/host/hashandcompress
L=$O/hostpath2.log; rm -f $L
for c in 1 2; do echo "== random16g -c $c skein512+lz4 64K" >> $L; $H -g true -c $c -r 1 --block-size=65536 -H skein512 -C lz4 /tmp/random16g.bin >> $L 2>&1; done
echo "== random16g -c 2 skein+lz4 4K" >> $L; $H -g true -c 2 -r 8 -H skein -C lz4 /tmp/random16g.bin >> $L 2>&1
echo "== random16g -c 2 sha256mb+lzf 4K" >> $L; $H -g true -c 2 -r 8 -H sha256mb -C lzf /tmp/random16g.bin >> $L 2>&1
for c in 1 2; do echo "== corpus4g -c $c skein512+lz4 64K" >> $L; $H -v -g true -c $c -r 1 --block-size=65536 -H skein512 -C lz4 /tmp/corpus4g.bin >> $L 2>&1; done
echo "== corpus4g -c 2 skein+lz4 4K" >> $L; $H -g true -c 2 -r 8 -H skein -C lz4 /tmp/corpus4g.bin >> $L 2>&1
echo "== corpus4g -c 2 sha256mb+lzf 4K" >> $L; $H -g true -c 2 -r 8 -H sha256mb -C lzf /tmp/corpus4g.bin >> $L 2>&1
echo "== corpus4g -g false -c 8 skein+lz4 4K (slots, first 64 MiB)" >> $L; head -c 67108864 /tmp/corpus4g.bin > /tmp/c64m.bin; $H -g false -c 8 -r 8 -H skein -C lz4 /tmp/c64m.bin >> $L 2>&1
cat $L
rm -f /tmp/random16g.bin /tmp/corpus4g.bin /tmp/c64m.bin
