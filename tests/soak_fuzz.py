#!/usr/bin/env python3
"""Time-boxed randomized soak of the codecs and decoders against the oracle (test infrastructure; run by hand on a GPU box):
    CW_LZ4_LANES=1 CW_LZF_LANES=1 CW_LZF_ROUND=7 CW_LANES_RESERVE=9 CW_DECODE_LANES=1 python tests/soak_fuzz.py 120   # lane kernels
    python tests/soak_fuzz.py 120                                                                                  # launch policy as is
Random block sizes (1 .. 65536), counts, strides of synthetic blocks (tests/test_gpu_fuzz.py::_block) until the time is up."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import compute_war_amd as cw  # noqa: E402
import oracle  # noqa: E402
from test_gpu_fuzz import _block  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
cw.init(0)
t0, rounds, blocks, bad = time.time(), 0, 0, 0
SIZES = [4096, 65536, 2048, 1024, 512, 8192, 16384, 32768, 65535, 4095, 4097, 20000, 333, 100, 17, 16, 13, 12, 5, 1]
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed0 + rounds)
    n = int(rng.choice(SIZES)) if rng.random() < 0.7 else int(rng.integers(1, 65537))
    count = int(rng.integers(1, max(2, min(400, (6 << 20) // n))))
    data = np.concatenate([_block(rng, n) if n >= 16 else rng.integers(0, 4, n, dtype=np.uint8) for _ in range(count)]).tobytes()
    for alg, comp in (("lz4", oracle.lz4_compress), ("lzf", oracle.lzf_compress)):
        sizes, payload = cw.compress_blocks(alg, data, n)
        for i in range(count):
            want = comp(data[i * n:(i + 1) * n])
            if sizes[i] != len(want) or payload[i, :sizes[i]].tobytes() != want:
                bad += 1
                print("MISMATCH", alg, "n", n, "block", i, "of", count, "seed", seed0 + rounds, int(sizes[i]), len(want), flush=True)
        out, status = cw.decompress_blocks(alg, sizes, payload, n)
        for i in range(count):
            if sizes[i] and (status[i] != 0 or out[i].tobytes() != data[i * n:(i + 1) * n]):
                bad += 1
                print("DECODE", alg, "n", n, "block", i, "seed", seed0 + rounds, flush=True)
    rounds += 1
    blocks += 2 * count
print(f"soak: {rounds} rounds, {blocks} block compressions + decodes against the oracle in {time.time() - t0:.0f} s, seed0 {seed0}: {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
