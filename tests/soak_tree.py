#!/usr/bin/env python3
"""Time-boxed randomized soak of Skein tree hashing (cw_hash_tree_blocks) against the oracle's tree restatement (test
infrastructure; run by hand on a GPU box):  python tests/soak_tree.py 60 [seed]
Random block sizes 0..65536, counts, leaf / node / max_level parameters; a parameter set the library rejects must be rejected
cleanly (CwError), everything it accepts must give the oracle's digest."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import compute_war_amd as cw  # noqa: E402
import oracle  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
cw.init(0)
t0, rounds, digests, bad, rejected = time.time(), 0, 0, 0, 0
SIZES = [65536, 4096, 1019, 64, 31, 0, 1, 32768, 8192, 65535, 100, 2048]
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed0 + rounds)
    bs = int(rng.choice(SIZES)) if rng.random() < 0.7 else int(rng.integers(0, 65537))
    count = int(rng.integers(1, max(2, min(80, (2 << 20) // max(bs, 1)))))
    alg, state_bits, hash_bits = (("skein512", 512, 512), ("skein", 256, 128))[rounds % 2]
    leaf, node = int(rng.integers(1, 8)), int(rng.integers(1, 5))
    ml = int(rng.choice([2, 3, 4, 255]))
    data = rng.integers(0, 256, max(bs * count, 1), dtype=np.uint8).tobytes()[:bs * count]
    if os.environ.get("CW_SOAK_VERBOSE"):
        print("round", rounds, alg, "bs", bs, "count", count, "leaf", leaf, "node", node, "ml", ml, flush=True)
    try:
        if bs == 0:
            raise cw.CwError(-1, "block size 0 goes through the device API in tests/soak_hash.py")
        dig = cw.hash_tree_blocks(alg, data, bs, leaf, node, ml)
    except cw.CwError:
        rejected += 1
        rounds += 1
        continue
    for i in sorted(set([0, count - 1] + [int(x) for x in rng.integers(0, count, min(count, 6))])):
        want = oracle.skein_tree(state_bits, data[i * bs:(i + 1) * bs], hash_bits, leaf, node, ml)
        digests += 1
        if dig[i].tobytes() != want:
            bad += 1
            print("MISMATCH", alg, "bs", bs, "block", i, "leaf", leaf, "node", node, "ml", ml, "seed", seed0 + rounds, flush=True)
    rounds += 1
print(f"soak: {rounds} rounds ({rejected} parameter sets rejected), {digests} tree digests against the oracle in {time.time() - t0:.0f} s, seed0 {seed0}: {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
