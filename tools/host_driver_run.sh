#!/bin/bash
# The C driver (compute_war_amd/host/hashandcompress, the reference's main() over the C ABI) on large files, as the reference's
# run_tests runs it: a 16 GiB file of noise and the in-tree corpora tiled to 4 GiB, in /dev/shm (reading is outside the timed
# window anyway).  Output: the reference's report lines  hash|comp|ms|MB/s  (+ totals with -v).   gpurun -- 'bash tools/host_driver_run.sh'
set -e
D=/dev/shm/cw_driver_$$; mkdir -p $D; trap "rm -rf $D" EXIT
EXE=compute_war_amd/host/hashandcompress
python3 - "$D" <<'PY'
import glob, os, sys
import numpy as np
d = sys.argv[1]
rng = np.random.default_rng(1)
with open(os.path.join(d, "random16g"), "wb") as f:
    for _ in range(64):
        f.write(rng.integers(0, 256, 256 << 20, dtype=np.uint8).tobytes())
tile = b"".join(open(p, "rb").read() for p in sorted(glob.glob("tests/golden/corpus/*/*")))
with open(os.path.join(d, "corpus4g"), "wb") as f:
    n = 0
    while n < (4 << 30):
        f.write(tile); n += len(tile)
PY
for cfg in "random16g 1 skein512 lz4 65536" "random16g 2 skein512 lz4 65536" "random16g 1 skein lz4 4096" "random16g 1 sha256mb lzf 4096" \
           "corpus4g 1 skein512 lz4 65536" "corpus4g 1 skein lz4 4096" "corpus4g 1 sha256mb lzf 4096"; do
  set -- $cfg
  echo "== $1 -c $2 $3+$4 $5 B blocks"
  # (CW_DRIVER_ALL_THREADS: the driver otherwise runs ONE worker per device on the offload path whatever -c says)
  CW_DRIVER_ALL_THREADS=1 $EXE -v --gpu-offload=true --c-threads=$2 --block-size=$5 --hash-alg=$3 --comp-alg=$4 $D/$1
done
