#!/usr/bin/env python3
"""Time-boxed randomized soak of the host batch entry points against the oracle (test infrastructure; run by hand on a GPU box):
    CW_HOST_CHUNK_MB=1 python tests/soak_host.py 120 [seed]     # 1 MiB chunks: many chunks, all three slots, partial last chunks
cw_hash_and_compress_blocks / cw_hash_and_compress_packed (pageable and page-locked buffers) from one to three host threads at
once, random block sizes and counts, every digest, size and payload byte against the oracle."""
import os
import sys
import threading
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
HASH = {"skein512": lambda b: oracle.skein512(b, 512), "skein": lambda b: oracle.skein256(b, 128), "sha256mb": oracle.sha256}
SIZES = [4096, 65536, 2048, 8192, 16384, 1000, 333, 65535, 4097, 64, 20000]
lock = threading.Lock()
tot = {"rounds": 0, "blocks": 0, "bad": 0}


def one(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.choice(SIZES))
    count = int(rng.integers(1, max(2, min(600, (5 << 20) // n))))
    halg = ["skein512", "skein", "sha256mb"][seed % 3]
    calg = ("lz4", "lzf")[(seed // 3) % 2]
    data = np.concatenate([_block(rng, n) for _ in range(count)])
    form = seed % 3
    if form == 0:
        dig, sizes, payload = cw.hash_and_compress_blocks(halg, calg, data, n)
        get = lambda i: payload[i, :sizes[i]].tobytes()
    else:
        dig, sizes, offs, packed = cw.hash_and_compress_packed(halg, calg, data, n, pinned=form == 2)
        get = lambda i: packed[int(offs[i]):int(offs[i]) + int(sizes[i])].tobytes()
    enc = oracle.lz4_compress if calg == "lz4" else oracle.lzf_compress
    bad = 0
    for i in range(count):
        blk = data[i * n:(i + 1) * n].tobytes()
        want = enc(blk)
        if dig[i].tobytes() != HASH[halg](blk) or int(sizes[i]) != len(want) or get(i) != want:
            bad += 1
            print("MISMATCH", halg, calg, "n", n, "block", i, "of", count, "form", form, "seed", seed, flush=True)
    with lock:
        tot["rounds"] += 1
        tot["blocks"] += count
        tot["bad"] += bad


def worker(tid, nthreads, t_end):
    k = 0
    while time.time() < t_end:
        one(seed0 + 1000003 * tid + k)
        k += 1


t0 = time.time()
phase = budget / 3
for nthreads in (1, 2, 3):
    t_end = time.time() + phase
    ts = [threading.Thread(target=worker, args=(t, nthreads, t_end)) for t in range(nthreads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
print(f"soak: {tot['rounds']} calls, {tot['blocks']} blocks (digest + size + payload) against the oracle in {time.time() - t0:.0f} s from 1, 2 and 3 "
      f"threads, seed0 {seed0}: {tot['bad']} mismatches", flush=True)
sys.exit(1 if tot["bad"] else 0)
